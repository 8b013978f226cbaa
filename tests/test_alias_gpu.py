"""-m gpu: a caller written against the REFERENCE's import names (vdetlib.vdet.track, vdetlib.vdet.video_det,
vdetlib.utils.cython_nms: /root/reference/vdet/track.py:13, vdet/video_det.py:11) runs unchanged on the build and
reproduces the goldens recorded from the reference (tests/golden/proto_golden.json.gz, nms_golden.npz)."""
import copy

import numpy as np
import pytest

import synth
from test_pipeline_gpu import _close, _py

pytestmark = pytest.mark.gpu


def test_unchanged_caller_tracking_and_vid_nms(proto_golden):
    # --- what a T-CNN script does, verbatim import lines ---
    from vdetlib.utils.protocol import tracks_proto_from_boxes, det_score
    from vdetlib.utils.common import options
    from vdetlib.vdet.track import greedily_track_from_det, greedily_track_from_raw_dets
    from vdetlib.vdet.video_det import apply_vid_nms
    case = synth.proto_case()
    vid, det, det_info = case['vid'], case['det'], case['det_info']
    g = proto_golden['greedy_track']
    trk = synth.make_stub_tracker(tracks_proto_from_boxes)
    for ci in (1, 2):
        opts = options({'max_tracks': 5, 'thres': 0.2, 'nms_thres': 0.3})
        out = greedily_track_from_det(vid, copy.deepcopy(det), trk, lambda d, ci=ci: det_score(d, ci), opts)
        _close(_py(out), g['plain_det_c%d' % ci])
    for ci in (1, 4):
        out = greedily_track_from_raw_dets(vid, det_info, trk, ci, options({'max_tracks': 4, 'thres': 0.5}))
        _close(_py(out), g['plain_raw_c%d' % ci])
    gv = proto_golden['apply_vid_nms']
    for ci in (1, 3):
        out = apply_vid_nms(copy.deepcopy(det), ci, thres=0.9)
        assert [d['hash'] for d in out['detections']] == gv[str(ci)]


def test_unchanged_caller_cython_nms(nms_golden):
    from vdetlib.utils.cython_nms import nms, vid_nms
    z, index = nms_golden
    done = 0
    for i, c in enumerate(index['nms']):
        if not (60 <= c['n'] <= 1000):
            continue
        d = synth.dets5(c['seed'], c['n'], c['frac'], c['degenerate'], c['kind'])
        assert nms(d, c['thresh']) == z['nms_%d' % i].tolist(), c
        done += 1
    assert done >= 20
    for i, c in enumerate(index['vid_nms'][:4]):
        d = synth.dets6(c['seed'], c['n'], c['n_frames'], c['frac'])
        if c['n'] == 0:
            d = np.zeros((0, 6), np.float32)
        assert vid_nms(d, thresh=c['thresh']) == z['vid_nms_%d' % i].tolist(), c
