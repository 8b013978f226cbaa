"""vdet_argsort_volume: the per-(frame, class) descending argsort (utils/nms.pyx:25, vdet/video_det.py:93) -- the LSD
radix kernel (default) and the equalised counting sort (csrc/binsort_kernels.hpp, VDET_BINSORT=1) with its per-column
fallback, against numpy on the distributions that stress the counting sort's sub-bin map: uniform, normal,
softmax-like, many octaves, exact ties, non-finite scores, sparse exponents."""
import os

import numpy as np
import pytest
import torch

from vdetlib_amd import ops, _lib

pytestmark = pytest.mark.gpu

_ctxs = {}


def ctx_for(binsort):
    """one context per sort flavour (the knob is read when the context is created)"""
    if binsort not in _ctxs:
        old = os.environ.get("VDET_BINSORT")
        os.environ["VDET_BINSORT"] = "1" if binsort else "0"
        try:
            _ctxs[binsort] = _lib.Context(torch.cuda.current_device())
        finally:
            if old is None:
                del os.environ["VDET_BINSORT"]
            else:
                os.environ["VDET_BINSORT"] = old
    return _ctxs[binsort]


def score_key(s):
    """sortable key of a float32 score (csrc/nms_kernels.hpp score_key): larger = earlier; -0.0 == +0.0; NaN first"""
    s = np.asarray(s, dtype=np.float32).copy()
    s[s == 0] = 0.0
    b = s.view(np.uint32)
    k = np.where(b & 0x80000000, ~b, b | 0x80000000).astype(np.uint32)
    k[np.isnan(s)] = 0xFFFFFFFF
    return k


def expected(scores_fbc, thr=None):
    """[F,B,C] -> order [F,C,B], ncand [F,C]: descending key, ties by descending index; non-candidates last"""
    F, B, C = scores_fbc.shape
    order = np.empty((F, C, B), dtype=np.int64)
    ncand = np.empty((F, C), dtype=np.int32)
    idx = np.arange(B)
    for f in range(F):
        for c in range(C):
            col = scores_fbc[f, :, c]
            k = score_key(col).astype(np.int64)
            if thr is not None:
                k[~(col > np.float32(thr))] = 0
            order[f, c] = np.lexsort((-idx, -k))
            ncand[f, c] = int((k != 0).sum())
    return order, ncand


def run(scores_fbc, thr=None, layout="FBC", binsort=True):
    t = torch.from_numpy(scores_fbc).cuda()
    if layout == "FCB":
        t = t.permute(0, 2, 1).contiguous()
    ctx = ctx_for(binsort)
    o, n = ops.argsort_volume(t, score_thresh=thr, layout=layout, ctx=ctx)
    nfail = ctx.query(9)
    return o.cpu().numpy().astype(np.int64) & 0xFFFF, n.cpu().numpy(), nfail


def make(kind, rng, F, B, C):
    if kind == "uniform":
        return rng.random((F, B, C), dtype=np.float32)
    if kind == "normal":
        return rng.standard_normal((F, B, C)).astype(np.float32)
    if kind == "softmax":                  # heavy mass near 0, a tail up to 1
        return np.exp(-rng.exponential(6.0, (F, B, C))).astype(np.float32)
    if kind == "octaves":                  # log-uniform over 60 octaves, both signs
        return (np.exp2(rng.uniform(-40, 20, (F, B, C))) * rng.choice([-1.0, 1.0], (F, B, C))).astype(np.float32)
    if kind == "dups":                     # 10 % exact duplicates of other entries (equal keys inside a bin: index order)
        s = rng.random((F, B, C), dtype=np.float32)
        for f in range(F):
            for c in range(C):
                src = rng.integers(0, B, B // 10)
                dst = rng.integers(0, B, B // 10)
                s[f, dst, c] = s[f, src, c]
        return s
    if kind == "nonfinite":
        s = rng.standard_normal((F, B, C)).astype(np.float32)
        m = rng.random((F, B, C))
        for val in (np.nan, np.inf, -np.inf):   # (equal keys share a bin, which takes at most 10: 4 of each kind per column)
            for f in range(F):
                for c in range(C):
                    s[f, rng.integers(0, B, 4), c] = val
        for f in range(F):
            for c in range(C):
                s[f, rng.integers(0, B, 2), c] = 0.0
                s[f, rng.integers(0, B, 2), c] = -0.0
        return s
    if kind == "sparse_exp":               # a handful of keys in far-away octaves whose mantissas differ in the last bits only
        s = rng.random((F, B, C), dtype=np.float32)
        for j in range(8):                 # (one bin takes at most 10 keys)
            s[:, j * 7, :] = np.float32(2.0 ** -40) * np.float32(1.0 + j * 2.0 ** -23)
            s[:, j * 7 + 1, :] = -np.float32(2.0 ** 30) * np.float32(1.0 + (j % 3) * 2.0 ** -23)
        return s
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "normal", "softmax", "octaves", "dups", "nonfinite", "sparse_exp"])
@pytest.mark.parametrize("B", [1500, 5000, 10000])
def test_counting_sort_matches_numpy(kind, B):
    rng = np.random.default_rng(sum(map(ord, kind)) * 100003 + B)
    s = make(kind, rng, 2, B, 5)
    o, n, nfail = run(s)
    eo, en = expected(s)
    assert np.array_equal(n, en)
    assert np.array_equal(o, eo)
    assert nfail == 0, "these columns are spreadable: the counting sort must have done them itself"
    o2, n2, nfail2 = run(s, binsort=False)          # the default: the LSD radix kernel
    assert np.array_equal(o2, eo) and np.array_equal(n2, en) and nfail2 == -1


@pytest.mark.parametrize("B", [1025, 4096, 4097, 10240, 10241, 17000])
def test_sizes_and_variants(B):
    rng = np.random.default_rng(B)
    s = make("normal", rng, 1, B, 3)
    o, n, nfail = run(s)
    eo, en = expected(s)
    assert np.array_equal(o, eo) and np.array_equal(n, en) and nfail == 0


def test_more_problems_than_resident_workgroups():
    """persistent workgroups claim problems from a counter: 3 x 250 columns > 2 workgroups x 256 CUs"""
    rng = np.random.default_rng(7)
    s = make("uniform", rng, 3, 2000, 250)
    o, n, nfail = run(s)
    eo, en = expected(s)
    assert np.array_equal(o, eo) and np.array_equal(n, en) and nfail == 0


def test_tied_columns_fall_back_to_the_radix_sort():
    rng = np.random.default_rng(11)
    F, B, C = 2, 6000, 6
    s = rng.random((F, B, C), dtype=np.float32)
    s[:, :, 0] = np.round(s[:, :, 0] * 64) / 64          # quantised: 65 distinct values
    s[:, :, 1] = 0.25                                     # all equal
    s[:, :, 2] *= 0.4                                     # (nothing else shares the octave of the runs below)
    s[0, :10, 2] = 0.5                                    # 10 equal keys: the largest bin the counting sort takes
    s[1, :11, 2] = 0.5                                    # 11: one too many
    s[0, :, 3] = -np.inf                                  # padded frames (vdetlib_amd.io) look like this
    o, n, nfail = run(s)
    eo, en = expected(s)
    assert np.array_equal(o, eo) and np.array_equal(n, en)
    assert nfail == 2 + 2 + 1 + 1          # quantised x2 frames, constant x2, the 11-run, the -inf column


def test_threshold_and_fcb_layout():
    rng = np.random.default_rng(13)
    s = make("uniform", rng, 2, 3000, 4)
    o, n, nfail = run(s, thr=0.4)
    eo, en = expected(s, thr=0.4)
    live = np.arange(3000)[None, None, :] < en[:, :, None]
    assert np.array_equal(n, en) and np.array_equal(np.where(live, o, -1), np.where(live, eo, -1))
    # the tail holds the non-candidates (any order)
    assert all(set(o[f, c, en[f, c]:]) == set(eo[f, c, en[f, c]:]) for f in range(2) for c in range(4))
    assert nfail == -1                     # thresholded columns go straight to the radix kernel
    o2, n2, nfail2 = run(s, layout="FCB")
    eo2, en2 = expected(s)
    assert np.array_equal(o2, eo2) and np.array_equal(n2, en2) and nfail2 == 0


def test_volume_nms_with_a_recorded_tie_order(oracle):
    """the volume-scale form of nms(dets, thresh, order): quantised scores, ties visited in ASCENDING index order (what a
    different argsort would leave) -- the lists come from argsort_volume, every tied run is reversed, and the walk follows
    them: survivors == the oracle's greedy loop fed the same order (utils/nms.pyx:26-66)."""
    import synth
    F, B, C = 2, 700, 3
    boxes, scores = synth.video(31, F, B, C)
    scores = (np.round(scores * 12) / 12).astype(np.float32)             # 13 distinct values: long tied runs
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    order, ncand = ops.argsort_volume(ts)
    o = order.cpu().numpy().astype(np.int64) & 0xFFFF
    rec = o.copy()
    for f in range(F):
        for c in range(C):
            col = scores[f, o[f, c], c]
            start = 0
            for q in range(1, B + 1):
                if q == B or col[q] != col[start]:
                    rec[f, c, start:q] = o[f, c, start:q][::-1]          # ties by ascending index
                    start = q
    assert not np.array_equal(rec, o)
    ki, kc = ops.nms_volume_ordered(tb, torch.from_numpy(rec.astype(np.int16)).cuda(), ncand, 0.3)
    for f in range(F):
        for c in range(C):
            d = np.hstack([boxes[f], scores[f, :, c:c + 1]]).astype(np.float32)
            want = oracle.nms(d, 0.3, order=rec[f, c])
            n = int(kc[f, c])
            assert n == len(want) and ki[f, c, :n].cpu().numpy().tolist() == want


def test_volume_nms_rejects_lists_out_of_range(oracle):
    """caller-supplied lists are checked before they are walked: a count above B or a box index >= B is a ValueError
    (VDET_EINVAL), never an out-of-bounds access; the other lists' results are still those of their own walks"""
    import synth
    F, B, C = 2, 600, 3
    boxes, scores = synth.video(37, F, B, C)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    order, ncand = ops.argsort_volume(ts)
    ki, kc = ops.nms_volume_ordered(tb, order, ncand, 0.3)                # sane lists: fine
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3)
    assert np.array_equal(kc.cpu().numpy(), wcnt)
    bad = order.clone()
    bad[1, 2, 17] = 20000                                                   # an index >= B (as uint16)
    with pytest.raises(ValueError):
        ops.nms_volume_ordered(tb, bad, ncand, 0.3)
    badn = ncand.clone()
    badn[0, 1] = B + 5
    with pytest.raises(ValueError):
        ops.nms_volume_ordered(tb, order, badn, 0.3)
    ki2, kc2 = ops.nms_volume_ordered(tb, order, ncand, 0.3)              # the context is usable afterwards
    assert torch.equal(kc, kc2)


@pytest.mark.parametrize("B", [300, 2000])
def test_volume_nms_takes_caller_lists_at_any_alignment(oracle, B):
    """the caller's lists of vdet_nms_volume_ordered may sit at any 2-byte aligned address: the lane-per-list walk of small
    frames reads four candidates per 8-byte load only from 8-byte aligned lists (B % 4 == 0), and falls back to single
    entries otherwise -- same survivors either way"""
    import synth
    F, C = 3, 4
    boxes, scores = synth.video(53 + B, F, B, C)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    order, ncand = ops.argsort_volume(ts)
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3)
    for shift in (0, 1, 2, 3):
        pool = torch.zeros(F * C * B + 8, dtype=order.dtype, device="cuda")
        view = pool[shift:shift + F * C * B].view(F, C, B)
        view.copy_(order)
        assert view.data_ptr() % 8 == (pool.data_ptr() + 2 * shift) % 8 and view.is_contiguous()
        ki, kc = ops.nms_volume_ordered(tb, view, ncand, 0.3)
        assert np.array_equal(kc.cpu().numpy(), wcnt) and np.array_equal(ki.cpu().numpy(), widx), shift
