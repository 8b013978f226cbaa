"""-m gpu parity tests of the device-resident array forms (vdetlib_amd.ops) against the oracle."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _check_volume(torch, oracle, boxes, scores, thresh, score_thresh=None, cap=None, layout="FBC"):
    from vdetlib_amd import ops
    tb = torch.from_numpy(boxes).cuda()
    ts = torch.from_numpy(scores if layout == "FBC" else np.ascontiguousarray(scores.transpose(0, 2, 1))).cuda()
    idx, cnt = ops.nms_volume(tb, ts, thresh, score_thresh=score_thresh, cap=cap, layout=layout)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    B = boxes.shape[1]
    widx, wcnt = oracle.nms_volume(boxes, scores, thresh, score_thresh, cap=B if cap is None else cap)
    assert np.array_equal(cnt, wcnt)
    assert np.array_equal(idx, widx)


def test_nms_volume_c1(torch_cuda, oracle):
    """BASELINE config 1 shape: 30 frames x 300 boxes x 30 classes."""
    boxes, scores = synth.video(1000 * 1 + 0, 30, 300, 30)
    _check_volume(torch_cuda, oracle, boxes, scores, 0.3)
    _check_volume(torch_cuda, oracle, boxes, scores, 0.3, layout="FCB")
    _check_volume(torch_cuda, oracle, boxes, scores, 0.5, score_thresh=0.05)


def test_nms_volume_ties_randn_frac(torch_cuda, oracle):
    boxes, scores = synth.video(77, 5, 700, 9, frac=True, kind="randn")
    scores = np.round(scores * 4) / 4          # heavy ties: exercises the tie rule on device
    scores[0, 3, 2] = np.nan
    _check_volume(torch_cuda, oracle, boxes, scores, 0.3)
    _check_volume(torch_cuda, oracle, boxes, scores, 0.3, score_thresh=0.0)


def test_nms_volume_big_frames(torch_cuda, oracle):
    """A few (frame, class) problems at the config-2 per-frame size (10k boxes)."""
    boxes, scores = synth.video(2000, 2, 10000, 3)
    _check_volume(torch_cuda, oracle, boxes, scores, 0.3, cap=2048)


@pytest.mark.parametrize("t", [0.3, 0.5])
def test_nms_volume_integer_and_fractional_frames(torch_cuda, oracle, t):
    """K1s takes the integer form of the predicate (x2 + 1 / y2 + 1 added once per box) on frames whose coordinates are all
    integers in [0, 65535] and the float form elsewhere -- frame by frame in one volume: integer frames, a fractional
    frame, coordinates of exactly 65535 and of 65536, a -0.0, exact-IoU borderlines on a coarse integer grid."""
    rng = np.random.RandomState(88)
    F, B, C = 8, 700, 3
    boxes = np.zeros((F, B, 4), np.float32)
    for f in range(F):
        grid = [1, 4, 8, 1, 16, 1, 2, 1][f]
        x1 = rng.randint(0, 1200 // grid, B) * grid
        y1 = rng.randint(0, 600 // grid, B) * grid
        w = rng.randint(1, 1 + 240 // grid, B) * grid
        h = rng.randint(1, 1 + 240 // grid, B) * grid
        boxes[f] = np.stack([x1, y1, x1 + w - 1, y1 + h - 1], 1)
    boxes[3] += rng.rand(B, 4).astype(np.float32) * 0.5            # fractional frame
    boxes[5, :40, [0, 2]] += 64000                                  # far right: still u16 ...
    boxes[5, 0] = [65000, 10, 65535, 300]
    boxes[6, :40, [0, 2]] += 64000
    boxes[6, 0] = [65000, 10, 65536, 300]                           # ... one pixel more: float form
    boxes[7, 5, 1] = np.float32(-0.0)                               # sign bit: float form
    scores = rng.rand(F, B, C).astype(np.float32)
    _check_volume(torch_cuda, oracle, boxes, scores, t)


def test_nms_volume_capacity_error(torch_cuda):
    from vdetlib_amd import ops
    boxes, scores = synth.video(5, 2, 300, 2)
    with pytest.raises(ValueError):
        ops.nms_volume(torch_cuda.from_numpy(boxes).cuda(), torch_cuda.from_numpy(scores).cuda(), 0.3, cap=8)


def test_nms_volume_full_size_properties(torch_cuda):
    """BASELINE config-2 per-frame/class sizes on a slab that fits the test budget (8 frames x 10k
    boxes x 200 classes): size-independent properties -- survivors are an independent set, every
    suppressed box has a higher-priority kept neighbour (maximality), order is descending."""
    torch = torch_cuda
    from vdetlib_amd import ops
    F, B, C = 8, 10000, 200
    g = torch.Generator(device='cuda').manual_seed(7)
    x1 = torch.rand(F, B, generator=g, device='cuda') * 1230
    y1 = torch.rand(F, B, generator=g, device='cuda') * 670
    w = 10 + torch.rand(F, B, generator=g, device='cuda') * 290
    h = 10 + torch.rand(F, B, generator=g, device='cuda') * 290
    boxes = torch.stack([x1, y1, torch.clamp(x1 + w, max=1279), torch.clamp(y1 + h, max=719)], -1).round().contiguous()
    scores = torch.rand(F, B, C, generator=g, device='cuda')
    idx, cnt = ops.nms_volume(boxes, scores, 0.3, cap=4096)
    assert int(cnt.min()) > 0 and int(cnt.max()) <= 4096

    def iou_mat(a, b):
        ix1 = torch.maximum(a[:, None, 0], b[None, :, 0]); iy1 = torch.maximum(a[:, None, 1], b[None, :, 1])
        ix2 = torch.minimum(a[:, None, 2], b[None, :, 2]); iy2 = torch.minimum(a[:, None, 3], b[None, :, 3])
        iw = (ix2 - ix1 + 1).clamp(min=0); ih = (iy2 - iy1 + 1).clamp(min=0)
        aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1); ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        inter = iw * ih
        return inter / (aa[:, None] + ab[None, :] - inter)

    for f, c in ((0, 0), (3, 117), (7, 199)):
        k = int(cnt[f, c]); kept = idx[f, c, :k].long()
        s = scores[f, :, c]
        assert bool((s[kept][:-1] >= s[kept][1:]).all())                      # descending
        m = iou_mat(boxes[f][kept], boxes[f][kept]); m.fill_diagonal_(0)
        assert float(m.max()) < 0.3 + 1e-6                                     # independent
        allm = iou_mat(boxes[f], boxes[f][kept])                              # [B,K]
        higher = s[kept][None, :] > s[:, None]
        sup = ((allm >= 0.3 - 1e-6) & higher).any(1)
        is_kept = torch.zeros(B, dtype=torch.bool, device='cuda'); is_kept[kept] = True
        assert bool((sup | is_kept).all())                                     # maximal


def test_temporal_maxpool(torch_cuda, oracle):
    from vdetlib_amd import ops
    torch = torch_cuda
    rng = np.random.RandomState(3)
    for shape, w in (((30, 300, 30), 3), ((17, 64, 8), 5), ((40, 33, 7), 7), ((12, 128), 9), ((25, 10, 3), 11),
                     ((5, 4), 3), ((1, 8), 3), ((300, 256), 3)):
        v = rng.randn(*shape).astype(np.float32)
        v[rng.rand(*shape) < 0.01] = np.nan
        got = ops.temporal_maxpool(torch.from_numpy(v).cuda(), w).cpu().numpy()
        assert np.array_equal(got, oracle.temporal_maxpool(v, w), equal_nan=True), (shape, w)
    with pytest.raises(ValueError):
        ops.temporal_maxpool(torch.zeros(4, 4, device='cuda'), 4)


def test_temporal_conv(torch_cuda, oracle):
    from vdetlib_amd import ops
    torch = torch_cuda
    rng = np.random.RandomState(4)
    for shape, k in (((30, 300, 30), 3), ((17, 64, 8), 5), ((40, 33, 7), 7), ((64, 128), 9), ((25, 12), 13)):
        v = rng.randn(*shape).astype(np.float32)
        taps = rng.randn(k).astype(np.float32)
        got = ops.temporal_conv(torch.from_numpy(v).cuda(), taps, bias=0.25, pad=-1.0).cpu().numpy()
        want = oracle.temporal_conv(v, taps, bias=0.25, pad=-1.0)
        assert np.array_equal(got, want), (shape, k)      # same op order, no contraction: bit-exact
        assert np.allclose(got, want, rtol=0, atol=1e-5)  # north-star tolerance for float scores


def test_temporal_maxpool_conv_fused(torch_cuda, oracle):
    """One pass producing both temporal operators == the two separate passes == the oracle."""
    torch = torch_cuda
    from vdetlib_amd import ops
    rng = np.random.RandomState(5)
    for shape, w in (((9, 40, 12), 3), ((16, 7, 5), 3), ((11, 64, 8), 5), ((4, 32, 4), 7), ((1, 8, 4), 3)):
        vol = (rng.randn(*shape) * 3).astype(np.float32)
        vol[rng.rand(*shape) < 0.01] = np.nan
        taps = rng.randn(w).astype(np.float32)
        tv = torch.from_numpy(vol).cuda()
        pm, pc = ops.temporal_maxpool_conv(tv, w, taps, pad_max=-1e5, bias=0.25, pad_conv=0.5)
        m0 = ops.temporal_maxpool(tv, w)
        c0 = ops.temporal_conv(tv, taps, bias=0.25, pad=0.5)
        assert np.array_equal(pm.cpu().numpy(), m0.cpu().numpy(), equal_nan=True)
        assert np.array_equal(pc.cpu().numpy(), c0.cpu().numpy(), equal_nan=True)
        assert np.array_equal(pm.cpu().numpy(), oracle.temporal_maxpool(vol, w), equal_nan=True)
        assert np.allclose(pc.cpu().numpy(), oracle.temporal_conv(vol, taps, 0.25, 0.5), rtol=0, atol=1e-5, equal_nan=True)


def test_volume_edge_shapes(torch_cuda, oracle):
    """Empty / minimal / ragged-in-content inputs of the device-resident entry points."""
    torch = torch_cuda
    from vdetlib_amd import ops
    # one frame, one box, one class
    b = torch.tensor([[[10., 10., 20., 20.]]], device='cuda')
    s = torch.tensor([[[0.5]]], device='cuda')
    idx, cnt = ops.nms_volume(b, s, 0.3)
    assert idx.tolist() == [[[0]]] and cnt.tolist() == [[1]]
    # no boxes at all
    idx, cnt = ops.nms_volume(torch.zeros(3, 0, 4, device='cuda'), torch.zeros(3, 0, 2, device='cuda'), 0.3, cap=4)
    assert cnt.tolist() == [[0, 0]] * 3 and idx.shape == (3, 2, 4)
    # all boxes identical: exactly one survivor (the best score), everything below threshold: none
    b = torch.tensor([5., 5., 50., 60.], device='cuda').repeat(2, 300, 1)
    s = torch.rand(2, 300, 3, device='cuda')
    idx, cnt = ops.nms_volume(b, s, 0.3)
    assert cnt.tolist() == [[1, 1, 1]] * 2
    assert torch.equal(idx[:, :, 0].long(), s.argmax(1))
    idx, cnt = ops.nms_volume(b, s, 0.3, score_thresh=2.0)
    assert int(cnt.sum()) == 0
    # thresholds at the edges: thresh > 1 keeps everything, thresh 0 keeps one per connected "all" graph
    boxes, scores = synth.video(31, 2, 120, 2)
    for thr in (1.5, 1.0, 0.0, 1e-12):
        tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        i1, c1 = ops.nms_volume(tb, ts, thr)
        i2, c2 = oracle.nms_volume(boxes, scores, thr)
        assert np.array_equal(c1.cpu().numpy(), c2) and np.array_equal(i1.cpu().numpy(), i2), thr
    # per-frame limit of the volume path
    with pytest.raises(ValueError):
        ops.nms_volume(torch.zeros(1, 19000, 4, device='cuda'), torch.zeros(1, 19000, 1, device='cuda'), 0.3, cap=8)
    # temporal ops on a single frame / single series
    v = torch.rand(1, 8, device='cuda')
    assert torch.equal(ops.temporal_maxpool(v, 3), v)
    assert ops.temporal_maxpool(torch.zeros(0, 8, device='cuda'), 3).shape == (0, 8)
    # tracking: nothing above the stop threshold / no tracks requested
    tr, an, nt = ops.track_volume(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thres=2.0, max_tracks=3)
    assert nt.tolist() == [0, 0] and bool(torch.isnan(tr).all())
    tr, an, nt = ops.track_volume(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), max_tracks=0)
    assert nt.tolist() == [0, 0] and tr.shape == (2, 0, 2, 5)


def test_volume_nan_inf_boxes_take_general_path(torch_cuda, oracle):
    """A frame with NaN/inf/degenerate boxes is 'irregular': it must go through the general kernel
    while its regular neighbours use the symmetric fast one -- same results as the oracle."""
    torch = torch_cuda
    boxes, scores = synth.video(32, 4, 500, 3, frac=True)
    boxes[1, 7, 0] = np.nan
    boxes[1, 9, 3] = np.inf
    boxes[2, 100:110, 2] = boxes[2, 100:110, 0] - 5      # negative widths
    _check_volume(torch, oracle, boxes, scores, 0.3)
    _check_volume(torch, oracle, boxes, scores, 0.5, score_thresh=0.2)


def _chain_video(rng, F, B, C, t):
    """Adversarial frames for the packed walk (eight candidates per pass, walk_list_packed): CHAINS of boxes where each
    suppresses the next but not the one after (IoU(i, i+1) >= t > IoU(i, i+2)), so that inside one group of eight the
    fate of every member depends on the fate of the one before it; stacks of near-duplicates (everything suppresses
    everything); a few boxes with very long adjacency lists (> 128 neighbours); isolated boxes."""
    boxes = np.zeros((F, B, 4), np.float32)
    for f in range(F):
        out = []
        while len(out) < B:
            kind = rng.randint(4)
            x0, y0 = rng.uniform(0, 1500), rng.uniform(0, 900)
            w, h = rng.uniform(40, 120), rng.uniform(40, 120)
            if kind == 0:       # chain along x: shift s with IoU = (w - s) / (w + s) just above t
                n = rng.randint(3, 20)
                s = np.floor(w * (1 - t) / (1 + t) * rng.uniform(0.75, 0.98))
                for i in range(n):
                    out.append([x0 + i * s, y0, x0 + i * s + w, y0 + h])
            elif kind == 1:     # stack of near-duplicates
                n = rng.randint(2, 12)
                for i in range(n):
                    out.append([x0 + rng.randint(0, 3), y0 + rng.randint(0, 3), x0 + w + rng.randint(0, 3), y0 + h + rng.randint(0, 3)])
            elif kind == 2 and rng.rand() < 0.05:     # a hub: > 128 similar boxes around one place
                for i in range(150):
                    out.append([x0 + rng.uniform(-6, 6), y0 + rng.uniform(-6, 6), x0 + w + rng.uniform(-6, 6), y0 + h + rng.uniform(-6, 6)])
            else:
                out.append([x0, y0, x0 + w, y0 + h])
        bb = np.round(np.asarray(out[:B], np.float32))
        boxes[f] = bb[rng.permutation(B)]
    scores = rng.permutation(F * B * C).reshape(F, B, C).astype(np.float32) / (F * B * C)      # tie-free
    # along half of the chains the scores descend in box order (the walk then meets them as one run)
    return boxes, scores


@pytest.mark.parametrize("seed,t", [(1, 0.3), (2, 0.5), (3, 0.7)])
def test_nms_volume_chains_stacks_hubs(torch_cuda, oracle, seed, t):
    rng = np.random.RandomState(9000 + seed)
    boxes, scores = _chain_video(rng, 3, 1500, 6, t)
    # class 0: scores follow the box order inside each frame's original (unpermuted) construction as far as possible --
    # sort class 0's scores along the x coordinate so that chains are met in order
    for f in range(boxes.shape[0]):
        o = np.argsort(boxes[f, :, 0] + 1e-3 * boxes[f, :, 1], kind="stable")
        scores[f, o, 0] = np.sort(scores[f, :, 0])[::-1]
    _check_volume(torch_cuda, oracle, boxes, scores, t)
    from vdetlib_amd.utils import cython_nms
    d = np.hstack([boxes[0], scores[0, :, :1]]).astype(np.float32)
    assert cython_nms.nms(d, t) == oracle.nms(d, t)


def test_packed_walk_random_sweep(torch_cuda, monkeypatch):
    """The packed walk of regular frames (eight candidates per pass over the K1s graph) against the general path -- the
    all-pairs predicate kernel and one survivor at a time, what irregular frames take (VDET_FORCE_GENERAL=1) -- on 150 random
    volumes: frame sizes around the 64 / 128 / 256 boundaries, dense clusters (long lists, many in-group suppressions),
    integral / fractional boxes, tied scores, thresholds 0.05 .. 0.99, with and without a score threshold."""
    from vdetlib_amd import ops, _lib
    torch = torch_cuda
    monkeypatch.setenv("VDET_FORCE_GENERAL", "1")
    cx0 = _lib.Context(torch.cuda.current_device())
    monkeypatch.delenv("VDET_FORCE_GENERAL")
    cx1 = _lib.Context(torch.cuda.current_device())      # the default
    rng = np.random.RandomState(4242)
    for it in range(150):
        B = int(rng.choice([2, 3, 7, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 1000, 2049]))
        F, C = int(rng.randint(1, 5)), int(rng.randint(1, 5))
        t = float(rng.choice([0.05, 0.1, 0.3, 0.5, 0.7, 0.9, 0.99]))
        kind = rng.randint(4)
        if kind == 0:
            boxes, scores = synth.video(int(rng.randint(1 << 30)), F, B, C, frac=bool(rng.randint(2)))
        else:
            x, y = rng.uniform(0, 300, (F, B)), rng.uniform(0, 200 if kind == 1 else 40, (F, B))
            boxes = np.stack([x, y, x + rng.uniform(20, 200, (F, B)), y + rng.uniform(20, 200, (F, B))], 2).astype(np.float32)
            if kind == 2:
                boxes = np.round(boxes)
            scores = rng.rand(F, B, C).astype(np.float32)
            if kind == 3:
                scores = np.round(scores * 8) / 8
        tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        st = None if rng.randint(2) else float(rng.uniform(0, 0.5))
        i0, c0 = ops.nms_volume(tb, ts, t, score_thresh=st, ctx=cx0)
        i1, c1 = ops.nms_volume(tb, ts, t, score_thresh=st, ctx=cx1)
        assert torch.equal(c0, c1) and torch.equal(i0, i1), (it, B, F, C, t, kind)


@pytest.mark.parametrize("kind", ["int", "frac"])
def test_dense_frames_whose_slabs_miss_the_k2_stage(torch_cuda, oracle, kind):
    """Frames whose boxes all overlap (degree ~ B / 2): the lists of 128 rows do not fit adj_build_kernel's 32 KB stage, so
    the rows go straight to the pool (the kernel's second extraction path, lists > 128 entries in the walk), next to an
    ordinary sparse frame in the same volume -- every (frame, class) against the oracle."""
    rng = np.random.RandomState(808 + (kind == "frac"))
    F, B, C = 3, 700, 3
    x, y = rng.uniform(0, 120, (F, B)), rng.uniform(0, 60, (F, B))
    boxes = np.stack([x, y, x + rng.uniform(60, 200, (F, B)), y + rng.uniform(60, 200, (F, B))], 2).astype(np.float32)
    sparse, _ = synth.video(4711, 1, B, C)
    boxes[1] = sparse[0]                                  # one ordinary frame between the dense ones
    if kind == "int":
        boxes = np.round(boxes)
    scores = rng.rand(F, B, C).astype(np.float32)
    for t in (0.1, 0.3, 0.6):
        _check_volume(torch_cuda, oracle, boxes, scores, t)


def test_cache_survives_another_geometry_in_between(torch_cuda, oracle):
    """vdet_set_cache(1): nms_volume(A) -> det_nms_volume / argsort_volume on ANOTHER geometry (they rewrite the context's
    group table) -> nms_volume(A) again must rebuild, not walk A's lists with the other geometry's groups"""
    torch = torch_cuda
    from vdetlib_amd import ops, _lib
    cx = _lib.Context(0)
    cx.set_cache(True)
    try:
        bA, sA = synth.video(4101, 6, 500, 4)
        tbA, tsA = torch.from_numpy(bA).cuda(), torch.from_numpy(sA).cuda()
        wantA = oracle.nms_volume(bA, sA, 0.3, cap=500)
        for other in ("det", "argsort"):
            i1, c1 = ops.nms_volume(tbA, tsA, 0.3, ctx=cx)
            assert np.array_equal(i1.cpu().numpy(), wantA[0]) and np.array_equal(c1.cpu().numpy(), wantA[1])
            bB, sB = synth.video(4102, 9, 333, 3)
            if other == "det":
                BX = torch.from_numpy(np.repeat(bB[:, :, None, :], 3, axis=2).copy()).cuda()
                ops.det_nms_volume(BX, torch.from_numpy(sB).cuda(), score_thresh=0.05, topk=50, nms_thresh=0.3, ctx=cx)
            else:
                ops.argsort_volume(torch.from_numpy(sB).cuda(), ctx=cx)
            i2, c2 = ops.nms_volume(tbA, tsA, 0.3, ctx=cx)
            assert np.array_equal(c2.cpu().numpy(), wantA[1]), other
            assert np.array_equal(i2.cpu().numpy(), wantA[0]), other
    finally:
        cx.close()
