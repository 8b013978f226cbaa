"""-m gpu: the Fast R-CNN per-class flow on the device (vdet_det_nms_volume, csrc/detnms_kernels.hpp): every class
suppresses its OWN regressed boxes -- fast_rcnn_det_vid's per-class loop (vdet/video_det.py:89-99) followed by
apply_image_nms (vdet/image_det.py:117-123) for every frame and class.  Rows against the golden the reference
recorded (proto_golden['fast_rcnn_det_vid']), kept lists against the pinned oracle's nms of those rows; then larger
random volumes and the config-2 size against the oracle's restatement of the same two steps."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _want(oracle, scores, boxes, thresh, k, nms_thresh):
    """oracle: rows per class of ONE frame (video_det.py:89-99) + keep list of each (image_det.py:117-123)"""
    rows = oracle.threshold_topk(scores, boxes, thresh, k)
    keeps = [[] if j == 0 else oracle.nms(np.ascontiguousarray(rows[j], dtype=np.float32).reshape(-1, 5), nms_thresh)
             for j in range(len(rows))]
    return rows, keeps


def _check_frame(oracle, f, S, BX, out, thresh, k, nms_thresh, classes=None):
    dets, sel, dcnt, keep, kcnt = out
    K = S.shape[2]
    rows, keeps = _want(oracle, S[f], BX[f].reshape(S.shape[1], 4 * K), thresh, k, nms_thresh)
    assert dcnt[f, 0] == 0 and kcnt[f, 0] == 0
    for j in (classes if classes is not None else range(1, K)):
        n = int(dcnt[f, j])
        want = np.asarray(rows[j], np.float32).reshape(-1, 5)
        assert n == len(want), (f, j, n, len(want))
        assert np.array_equal(dets[f, j, :n], want), (f, j)
        assert np.array_equal(BX[f].reshape(-1, K, 4)[sel[f, j, :n], j], want[:, :4]), (f, j)
        assert keep[f, j, :int(kcnt[f, j])].tolist() == keeps[j], (f, j)
        assert np.all(keep[f, j, int(kcnt[f, j]):] == -1)


def test_fast_rcnn_flow_against_the_reference_rows(oracle, proto_golden):
    import torch
    from vdetlib_amd import ops
    g = proto_golden['fast_rcnn_det_vid']
    Fv, Bv, Cv = g['F'], g['B'], g['C']
    box6 = synth.make_box_proto(g['box_seed'], 'synth_vid_b', Fv, Bv)
    det_fun = synth.det_fun_case(Cv)
    S = np.zeros((Fv, Bv, Cv + 1), np.float32)
    BX = np.zeros((Fv, Bv, 4 * (Cv + 1)), np.float32)
    for f in range(Fv):
        prop = np.array([b['bbox'] for b in box6['boxes'] if b['frame'] == f + 1])
        sc, bx = det_fun(None, None, prop)
        S[f], BX[f] = sc.astype(np.float32), bx.astype(np.float32)
    tS, tB = torch.from_numpy(S).cuda(), torch.from_numpy(BX).cuda()
    for key, k, thr in (('full', 100, 0.05), ('top20', 20, 0.5)):
        out = [t.cpu().numpy() for t in ops.det_nms_volume(tB, tS, score_thresh=thr, topk=k, nms_thresh=0.3)]
        dets, sel, dcnt, keep, kcnt = out
        for j in range(1, Cv + 1):
            for i in range(Fv):
                want = np.asarray(g[key][j][i], dtype=np.float32).reshape(-1, 5)      # what the REFERENCE returned
                n = int(dcnt[i, j])
                assert n == len(want) and np.array_equal(dets[i, j, :n], want), (key, j, i)
                assert keep[i, j, :int(kcnt[i, j])].tolist() == oracle.nms(want, 0.3), (key, j, i)
        assert np.all(dcnt[:, 0] == 0)


@pytest.mark.parametrize("cfg", [dict(seed=5101, F=3, B=700, K=6, thr=0.05, k=100, frac=False),
                                 dict(seed=5102, F=2, B=90, K=4, thr=0.3, k=100, frac=True),        # fewer candidates than the cut: box order
                                 dict(seed=5103, F=2, B=1500, K=3, thr=None, k=128, frac=True),
                                 dict(seed=5104, F=4, B=257, K=9, thr=0.6, k=1, frac=False),
                                 dict(seed=5105, F=2, B=64, K=3, thr=0.99, k=50, frac=False)])     # almost nothing passes
def test_det_nms_volume_vs_oracle(oracle, cfg):
    import torch
    from vdetlib_amd import ops
    rng = np.random.RandomState(cfg['seed'])
    F, B, K = cfg['F'], cfg['B'], cfg['K']
    base = np.stack([synth.boxes_1(rng, B, cfg['frac']) for _ in range(F)], 0)                      # [F,B,4]
    BX = (base[:, :, None, :] + rng.uniform(-12, 12, (F, B, K, 4))).astype(np.float32)
    if not cfg['frac']:
        BX = np.round(BX)
    BX[..., 2:] = np.maximum(BX[..., 2:], BX[..., :2])
    S = rng.rand(F, B, K).astype(np.float32)
    thr = cfg['thr']
    out = [t.cpu().numpy() for t in ops.det_nms_volume(torch.from_numpy(BX).cuda(), torch.from_numpy(S).cuda(),
                                                       score_thresh=thr, topk=cfg['k'], nms_thresh=0.3)]
    for f in range(F):
        _check_frame(oracle, f, S, BX, out, -np.inf if thr is None else thr, cfg['k'], 0.3)


def test_det_nms_tied_scores_beyond_the_cut(oracle):
    """quantised scores (many ties, more candidates than topk): after the cut the rows are argsort(-score)[:k] with equal
    scores by ASCENDING box index (vdet/video_det.py:93-97, stable), the NMS of those rows visits ties by descending row"""
    import torch
    from vdetlib_amd import ops
    rng = np.random.RandomState(5190)
    F, B, K = 3, 400, 5
    base = np.stack([synth.boxes_1(rng, B, False) for _ in range(F)], 0)
    BX = np.round(base[:, :, None, :] + rng.uniform(-12, 12, (F, B, K, 4))).astype(np.float32)
    BX[..., 2:] = np.maximum(BX[..., 2:], BX[..., :2])
    S = (np.floor(rng.rand(F, B, K) * 12) / 12).astype(np.float32)          # 12 levels: runs of ~33 equal scores
    for k in (100, 37, 128):
        out = [t.cpu().numpy() for t in ops.det_nms_volume(torch.from_numpy(BX).cuda(), torch.from_numpy(S).cuda(),
                                                           score_thresh=0.05, topk=k, nms_thresh=0.3)]
        for f in range(F):
            _check_frame(oracle, f, S, BX, out, 0.05, k, 0.3)


def test_det_nms_zero_union_raises_like_the_reference(oracle):
    """two identical zero-area boxes at the top of a class: the reference divides by a zero union (ZeroDivisionError)"""
    import torch
    from vdetlib_amd import ops
    rng = np.random.RandomState(77)
    F, B, K = 1, 40, 3
    BX = np.round(synth.boxes_1(rng, B)[None, :, None, :] + np.zeros((F, B, K, 4))).astype(np.float32)
    S = rng.rand(F, B, K).astype(np.float32)
    BX[0, 3, 2] = BX[0, 9, 2] = np.array([10, 10, 9, 30], np.float32)        # width 0 under the +1 convention
    S[0, 3, 2], S[0, 9, 2] = 2.0, 1.5
    rows = np.hstack([BX[0, :, 2], S[0, :, 2, None]]).astype(np.float32)
    with pytest.raises(ZeroDivisionError):
        oracle.nms(rows, 0.3)
    with pytest.raises(ZeroDivisionError):
        ops.det_nms_volume(torch.from_numpy(BX).cuda(), torch.from_numpy(S).cuda(), score_thresh=None, topk=100)
    with pytest.raises(ValueError):
        ops.det_nms_volume(torch.from_numpy(BX).cuda(), torch.from_numpy(S).cuda(), topk=129)


def test_det_nms_config2_size(oracle):
    """300 frames x 10 000 proposals x 200 classes (+ background), top-100 per (frame, class): 64 sampled problems against
    the oracle; whole-volume properties (counts, kept rows are rows, descending scores)"""
    import torch
    from vdetlib_amd import ops
    F, B, K = 300, 10000, 201
    g = torch.Generator(device="cuda").manual_seed(52)
    x1 = torch.rand(F, B, 1, generator=g, device="cuda") * 1230
    y1 = torch.rand(F, B, 1, generator=g, device="cuda") * 670
    w = 10 + torch.rand(F, B, 1, generator=g, device="cuda") * 290
    h = 10 + torch.rand(F, B, 1, generator=g, device="cuda") * 290
    BX = torch.empty(F, B, K, 4, device="cuda")
    for i, t in enumerate((x1, y1, x1 + w, y1 + h)):                                              # per-class regression deltas
        BX[..., i] = t + (torch.rand(F, B, K, generator=g, device="cuda") - 0.5) * 8       # (w, h >= 10: no box collapses)
    del x1, y1, w, h
    BX.round_()
    S = torch.rand(F, B, K, generator=g, device="cuda")
    dets, sel, dcnt, keep, kcnt = ops.det_nms_volume(BX, S, score_thresh=0.05, topk=100, nms_thresh=0.3)
    assert int(dcnt[:, 0].sum()) == 0 and bool((dcnt[:, 1:] == 100).all())
    assert bool((kcnt[:, 1:] > 0).all()) and bool((kcnt <= dcnt).all())
    sc = dets[..., 4]
    assert bool((sc[:, 1:, 1:] <= sc[:, 1:, :-1]).all())                                          # rows after the cut: descending score
    rng = np.random.RandomState(9)
    out = None
    for f in sorted(set(rng.randint(0, F, 8).tolist())):
        Sf, Bf = S[f].cpu().numpy(), BX[f].cpu().numpy()
        classes = sorted(set(rng.randint(1, K, 8).tolist()))
        o = [t[f:f + 1].cpu().numpy() for t in (dets, sel, dcnt, keep, kcnt)]
        _check_frame(oracle, 0, Sf[None], Bf[None], o, 0.05, 100, 0.3, classes=classes)
