"""CPU: host logic of the dict API that sits above the C-ABI -- checked with the ORACLE standing in for the device call (the
oracle is the checker here; the product never imports it):
 * the batching of the reference's per-tracked-box `track_det_nms` pattern (vdet/track.py:236-250) in
   vdetlib_amd/vdet/track.py::_prune_frame_dets reproduces the reference's recorded greedy tracks (proto_golden, incl. a tracker
   that returns several tracklets whose frames repeat);
 * the row builder of `apply_vid_nms` (C helper csrc/protofast.c and its itertools form) == `det_score` per detection, also
   for protos whose score lists are NOT in class order (first match wins, utils/protocol.py:323-327);
 * utils/timer.py keeps `average_time` a plain attribute (reference utils/timer.py:10-32)."""
import copy
import pickle

import numpy as np
import pytest

import synth
from test_pipeline_gpu import _close, _py


def _oracle_batch(oracle):
    def track_det_nms_batch(tracks, dets, offsets, thresh, track_offsets=None):
        assert track_offsets is None and len(tracks) == len(offsets) - 1
        keep = np.zeros(max(len(dets), 1), dtype=np.int64)
        counts = np.zeros(len(tracks), dtype=np.int64)
        for k in range(len(tracks)):
            kp = oracle.track_det_nms(np.ascontiguousarray(tracks[k:k + 1]), np.ascontiguousarray(dets[offsets[k]:offsets[k + 1]]), thresh)
            counts[k] = len(kp)
            keep[offsets[k]:offsets[k] + len(kp)] = kp
        return keep, counts
    return track_det_nms_batch


def test_batched_prune_reproduces_the_reference_tracks(oracle, proto_golden, monkeypatch):
    from vdetlib_amd.vdet import track as K
    from vdetlib_amd.utils import protocol as P, common as Cm
    calls = []
    fake = _oracle_batch(oracle)
    monkeypatch.setattr(K, 'track_det_nms_batch', lambda *a, **k: (calls.append(len(a[0])), fake(*a, **k))[1])
    case = synth.proto_case()
    g = proto_golden['greedy_track']
    vid, det, det_info = case['vid'], case['det'], case['det_info']
    for tag, kw in (('plain', {}), ('nan_split', {'nan_at': 1})):
        trk = synth.make_stub_tracker(P.tracks_proto_from_boxes, **kw)
        for ci in (1, 2):
            opts = Cm.options({'max_tracks': 5, 'thres': 0.2, 'nms_thres': 0.3})
            out = K.greedily_track_from_det(vid, copy.deepcopy(det), trk, lambda d, ci=ci: P.det_score(d, ci), opts)
            _close(_py(out), g['%s_det_c%d' % (tag, ci)])
        for ci in (1, 4):
            opts = Cm.options({'max_tracks': 4, 'thres': 0.5})
            out = K.greedily_track_from_raw_dets(vid, det_info, trk, ci, opts)
            _close(_py(out), g['%s_raw_c%d' % (tag, ci)])
    assert calls and max(calls) > 1          # several boxes of a tracklet really went out in one call


def test_prune_flushes_when_a_frame_repeats(oracle, monkeypatch):
    """two tracklets over the SAME frames: the second visit of a frame must see the first visit's result (the reference's
    sequential loop), so the batch is cut there -- compared with the one-call-per-box loop"""
    from vdetlib_amd.vdet import track as K
    fake = _oracle_batch(oracle)
    sizes = []
    monkeypatch.setattr(K, 'track_det_nms_batch', lambda *a, **k: (sizes.append(len(a[0])), fake(*a, **k))[1])
    rng = np.random.RandomState(5)
    n = 400
    rows = np.zeros((n, 6), np.float32)
    rows[:, 0] = rng.randint(1, 5, n)
    rows[:, 1:3] = rng.randint(0, 200, (n, 2))
    rows[:, 3:5] = rows[:, 1:3] + rng.randint(20, 120, (n, 2))
    rows[:, 5] = np.sort(rng.rand(n).astype(np.float32))[::-1]
    tracklets = [[{'frame': f, 'bbox': [10 * t + 20, 30, 10 * t + 140, 150]} for f in (1, 2, 3, 4, 9)] for t in range(3)]
    tracklets[1].append({'frame': 2, 'bbox': [100, 100, 180, 190]})        # a frame twice inside one tracklet
    rof = K._rows_by_frame(rows[:, 0])
    alive = np.ones(n, bool)
    K._prune_frame_dets(tracklets, rof, alive, rows, 0.3)
    want = np.ones(n, bool)
    for tr in tracklets:                     # the reference's loop, :236-250
        for box in tr:
            ids = [i for i in np.flatnonzero(rows[:, 0] == box['frame']) if want[i]]
            if not ids:
                continue
            kp = set(oracle.track_det_nms(np.asarray([[box['frame']] + box['bbox']], np.float32), rows[ids], 0.3))
            for q, i in enumerate(ids):
                if q not in kp:
                    want[i] = False
    assert np.array_equal(alive, want) and not alive.all()
    assert sizes == [4, 2, 4] or sum(sizes) == 4 * 3 + 1, sizes       # cut at every repeated frame (frame 9 has no detections)


def test_rows_by_frame_matches_a_plain_grouping():
    from vdetlib_amd.vdet import track as K
    keys = np.asarray([3, 1, 3, 2, 1, 3], np.float32)
    got = K._rows_by_frame(keys)
    assert sorted(got) == [1.0, 2.0, 3.0] and got[3].tolist() == [0, 2, 5] and got[1].tolist() == [1, 4]
    assert got.get(3) is got[3.0]                                     # an int frame id of a tracklet finds the float key
    odd = K._rows_by_frame(['a', 'b', 'a'])
    assert odd['a'].tolist() == [0, 2]


@pytest.mark.parametrize("use_c", [True, False])
def test_vid_nms_rows_equal_det_score(use_c, monkeypatch):
    from vdetlib_amd.vdet import video_det as V
    from vdetlib_amd.utils.protocol import det_score
    if use_c and V._protofast is None:
        pytest.skip("csrc/protofast.c not built")
    if not use_c:
        monkeypatch.setattr(V, '_protofast', None)
    det = synth.proto_case()['det']
    dets = copy.deepcopy(det['detections'])
    dets[3]['scores'][1]['class_index'] = 4          # an EARLIER entry claims class 4: first match wins in the reference
    dets[5]['scores'] = dets[5]['scores'][:2]        # a short list
    dets[6]['scores'] = list(reversed(dets[6]['scores']))
    for ci in (0, 1, 4, 5, 17, -1, np.int64(2)):
        rows = V._vid_nms_rows(dets, ci)
        want = np.asarray([[d['frame']] + list(d['bbox']) + [det_score(d, ci)] for d in dets], dtype=np.float32)
        assert rows.dtype == np.float32 and np.array_equal(rows, want), ci
    assert V._vid_nms_rows([], 1).shape == (0, 6)
    broken = copy.deepcopy(dets[:3])
    del broken[1]['bbox']
    with pytest.raises(KeyError):
        V._vid_nms_rows(broken, 1)


def test_timer_average_time_is_a_plain_attribute():
    from vdetlib_amd.utils.timer import Timer
    t = Timer()
    t.tic(); t.toc()
    assert 'average_time' in t.__dict__ and t.average_time == t.total_time / t.calls
    t.average_time = 7.5                              # scripts may assign it (reference: a plain attribute)
    assert t.average_time == 7.5
    u = pickle.loads(pickle.dumps(t))
    assert u.average_time == 7.5 and u.calls == 1
    t.reset()
    assert t.average_time == 0.0 and t.calls == 0
