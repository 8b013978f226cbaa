"""The LINK-stage oracle (oracle.greedy_track_volume / rescored_tubelets) against outputs of THE REFERENCE:
tests/golden/link_golden.npz holds what the reference's own greedily_track_from_raw_dets
(vdet/track.py:189-252), raw_dets_spatial_max_pooling + do_score_completion (vdet/tubelet_cls.py:493-535,
:284-303) and score_proto_temporal_maxpool (:386-414) produced when driven with the build's IoU-linking
tracker as ``track_method`` (tests/golden/make_golden.py g14)."""
import os

import numpy as np
import pytest

import synth
from conftest import GOLDEN


@pytest.fixture(scope="module")
def link_golden():
    return np.load(os.path.join(GOLDEN, 'link_golden.npz'))


def check_against_golden(z, case, tracks, ntracks, anchors, det, pooled, boxes):
    """Shared with the -m gpu test: tracks [C,T,F,5], ntracks [C], anchors [C,T,3] or None, det / pooled
    [C,T,F] f64 or None, boxes [C,T,F,4] -- all numpy."""
    n = 'link_' + case['name']
    want_nt = z[n + '_ntracks']
    assert np.array_equal(np.asarray(ntracks), want_nt), (case['name'], ntracks, want_nt)
    for c in range(len(want_nt)):
        k = int(want_nt[c])
        assert np.array_equal(tracks[c, :k], z[n + '_tracks'][c, :k], equal_nan=True), (case['name'], c)
        if anchors is not None:
            assert np.array_equal(anchors[c, :k, 0].astype(np.int32), z[n + '_anchor_frames'][c, :k]), (case['name'], c)
        assert np.array_equal(boxes[c, :k], z[n + '_boxes'][c, :k], equal_nan=True), (case['name'], c)
        # float scores: the north star's tolerance is 1e-5; completion / max-pool are exact here, keep 1e-9
        if det is not None:
            np.testing.assert_allclose(det[c, :k], z[n + '_det'][c, :k], rtol=0, atol=1e-9, equal_nan=True)
        np.testing.assert_allclose(pooled[c, :k], z[n + '_pooled'][c, :k], rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("case", synth.LINK_CASES, ids=[c['name'] for c in synth.LINK_CASES])
def test_oracle_link_chain_vs_reference(oracle, link_golden, case):
    boxes, scores = synth.link_case_video(case)
    C = scores.shape[2]
    wtr, wnt, wsc, wbx, wdet = oracle.rescored_tubelets(boxes, scores, case['nms_thres'], case['thres'], case['max_tracks'],
                                                        case['link'], case['pool'], case['window'], case['max_frames'],
                                                        return_det=True)
    anchors = np.zeros((C, case['max_tracks'], 3), np.float32)
    for c in range(C):
        _, a_, _ = oracle.greedy_track_volume(boxes, scores[:, :, c], case['nms_thres'], case['thres'], case['max_tracks'],
                                              case['link'], case['max_frames'])
        anchors[c] = a_
    check_against_golden(link_golden, case, wtr, wnt, anchors, wdet, wsc, wbx)


def test_link_golden_is_nontrivial(link_golden):
    """The cases do exercise what they claim: tracks longer than one frame, the thres stop, max_frames."""
    z = link_golden
    assert (np.sum(~np.isnan(z['link_plain_tracks'][..., 0]), axis=2) > 1).any()
    cs = [c for c in synth.LINK_CASES if c['name'] == 'thres_stop'][0]
    assert (z['link_thres_stop_ntracks'] < cs['max_tracks']).all()
    cm = [c for c in synth.LINK_CASES if c['name'] == 'max_frames'][0]
    assert np.sum(~np.isnan(z['link_max_frames_tracks'][..., 0]), axis=2).max() <= cm['max_frames']
    assert (z['link_plain_det'] != z['link_plain_pooled'])[~np.isnan(z['link_plain_det'])].any()
