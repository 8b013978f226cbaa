"""load_frame_to_det / load_det_info (utils/protocol.py:528-555) on per-frame .mat files against what the REFERENCE
returned for the same files (tests/golden/mat_golden.npz, written by make_golden.py --mat-only): missing frame file,
empty frame, both file-name conventions."""
import os

import numpy as np

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mat_golden.npz')


def test_mat_loaders_match_the_reference(tmp_path):
    from vdetlib_amd.utils import protocol as P
    z = np.load(GOLDEN)
    vid = synth.write_mat_case(str(tmp_path))
    ftd = P.load_frame_to_det(vid, str(tmp_path))
    assert sorted(ftd) == z['frames'].tolist()
    for f, (boxes, zs) in ftd.items():
        assert boxes.dtype == z['boxes_%d' % f].dtype and boxes.shape == z['boxes_%d' % f].shape
        assert np.array_equal(boxes, z['boxes_%d' % f]) and np.array_equal(zs, z['zs_%d' % f])
    info = P.load_det_info(vid, str(tmp_path))
    assert info.dtype == z['det_info'].dtype and info.shape == z['det_info'].shape
    assert np.array_equal(info, z['det_info'])
    # rows are [frame_id, x1, y1, x2, y2, scores...] in frame order; the empty frame (2) and the missing one (4) leave no rows
    assert sorted(set(info[:, 0].astype(int).tolist())) == [1, 3, 5, 6, 7]


def test_mat_files_feed_the_array_transport(tmp_path):
    """the same files through vdetlib_amd.io (ragged frames -> padded arrays): every real box and score arrives unchanged"""
    from vdetlib_amd.utils import protocol as P
    from vdetlib_amd import io as vio
    vid = synth.write_mat_case(str(tmp_path))
    ftd = P.load_frame_to_det(vid, str(tmp_path))
    boxes, scores, counts = vio.arrays_from_frame_to_det(vid, ftd)
    assert boxes.shape[0] == len(vid['frames']) and boxes.shape[2] == 4 and scores.shape[2] == synth.MAT_CASE['C']
    for i, frame in enumerate(vid['frames']):
        f = frame['frame']
        n = ftd[f][0].shape[0] if f in ftd else 0
        assert counts[i] == n
        if n:
            assert np.array_equal(boxes[i, :n], ftd[f][0].astype(np.float32))
            assert np.array_equal(scores[i, :n], ftd[f][1].astype(np.float32))
        assert np.all(np.isneginf(scores[i, n:]))
