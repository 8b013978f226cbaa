"""-m gpu: frames of at most 384 boxes (csrc/small_kernels.hpp: one wave sorts a list by counting, one block walks a frame's
classes against the frame's rows in LDS) -- the ILSVRC-VID shape.  The argsort against numpy (ties, duplicates, thresholds,
non-finite scores, every size around the 64-key boundaries), the NMS against the oracle, and both against the large-list
kernels (VDET_SMALL_LISTS=0), including irregular frames (left to the general walk) inside the same volume."""
import os

import numpy as np
import pytest
import torch

import synth
from test_argsort_gpu import expected, make

from vdetlib_amd import ops, _lib

pytestmark = pytest.mark.gpu

_ctxs = {}


def ctx_for(small):
    if small not in _ctxs:
        old = os.environ.get("VDET_SMALL_LISTS")
        os.environ["VDET_SMALL_LISTS"] = "1" if small else "0"
        try:
            _ctxs[small] = _lib.Context(torch.cuda.current_device())
        finally:
            if old is None:
                del os.environ["VDET_SMALL_LISTS"]
            else:
                os.environ["VDET_SMALL_LISTS"] = old
    return _ctxs[small]


def _argsort(s, thr=None, layout="FBC", small=True):
    t = torch.from_numpy(s).cuda()
    if layout == "FCB":
        t = t.permute(0, 2, 1).contiguous()
    o, n = ops.argsort_volume(t, score_thresh=thr, layout=layout, ctx=ctx_for(small))
    return o.cpu().numpy().astype(np.int64) & 0xFFFF, n.cpu().numpy()


@pytest.mark.parametrize("kind", ["uniform", "normal", "dups", "nonfinite", "octaves"])
@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 128, 129, 255, 256, 300, 320, 383, 384, 385])
def test_small_argsort_matches_numpy(kind, B):
    rng = np.random.default_rng(sum(map(ord, kind)) * 7919 + B)
    s = make(kind, rng, 3, B, 7) if B >= 20 or kind in ("uniform", "normal") else rng.random((3, B, 7), dtype=np.float32)
    eo, en = expected(s)
    for small in (True, False):
        o, n = _argsort(s, small=small)
        assert np.array_equal(n, en) and np.array_equal(o, eo), (kind, B, small)


def test_small_argsort_quantised_thresholds_layouts():
    """every list full of ties (scores on a grid of 16 values), a threshold that excludes about half, both layouts:
    candidates in descending score / descending index, the excluded tail by descending index"""
    rng = np.random.default_rng(5)
    for B in (40, 200, 300, 384):
        s = (np.round(rng.random((4, B, 6), dtype=np.float32) * 16) / 16).astype(np.float32)
        s[0, :, 0] = 0.25                                   # one list of ONE value
        s[1, :, 1] = np.nan
        for thr in (None, 0.5, 2.0, -1.0):
            eo, en = expected(s, thr)
            for layout in ("FBC", "FCB"):
                o, n = _argsort(s, thr, layout)
                assert np.array_equal(n, en), (B, thr, layout)
                assert np.array_equal(o, eo), (B, thr, layout)


def _nms_both(boxes, scores, t, score_thresh=None, layout="FBC"):
    tb = torch.from_numpy(boxes).cuda()
    ts = torch.from_numpy(scores if layout == "FBC" else np.ascontiguousarray(scores.transpose(0, 2, 1))).cuda()
    res = []
    for small in (True, False):
        idx, cnt = ops.nms_volume(tb, ts, t, score_thresh=score_thresh, layout=layout, ctx=ctx_for(small))
        res.append((idx.cpu().numpy(), cnt.cpu().numpy()))
    return res


@pytest.mark.parametrize("B", [1, 2, 64, 65, 130, 256, 300, 384])
def test_small_nms_vs_oracle(oracle, B):
    rng = np.random.RandomState(900 + B)
    F, C = 5, 9
    for frac in (False, True):
        boxes, scores = synth.video(int(rng.randint(1 << 30)), F, B, C, frac=frac)
        for t in (0.05, 0.3, 0.7):
            widx, wcnt = oracle.nms_volume(boxes, scores, t, None, cap=B)
            for (idx, cnt) in _nms_both(boxes, scores, t):
                assert np.array_equal(cnt, wcnt) and np.array_equal(idx, widx), (B, frac, t)


def test_small_nms_dense_tied_thresholded(oracle):
    """piled-up boxes (rows of > 100 bits, many in-chunk suppressions), scores on a grid (the tie path of the sort), a score
    threshold, both layouts, a cap smaller than the survivors"""
    rng = np.random.RandomState(77)
    F, B, C = 4, 333, 5
    x, y = rng.uniform(0, 150, (F, B)), rng.uniform(0, 60, (F, B))
    boxes = np.round(np.stack([x, y, x + rng.uniform(20, 160, (F, B)), y + rng.uniform(20, 160, (F, B))], 2)).astype(np.float32)
    scores = (np.round(rng.rand(F, B, C) * 32) / 32).astype(np.float32)
    for t in (0.1, 0.5, 0.9):
        for st in (None, 0.4):
            widx, wcnt = oracle.nms_volume(boxes, scores, t, st, cap=B)
            for layout in ("FBC", "FCB"):
                for (idx, cnt) in _nms_both(boxes, scores, t, st, layout):
                    assert np.array_equal(cnt, wcnt) and np.array_equal(idx, widx), (t, st, layout)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.9, None, cap=B)
    assert wcnt.max() > 20
    with pytest.raises(Exception):
        ops.nms_volume(tb, ts, 0.9, cap=20, ctx=ctx_for(True))


def test_irregular_frames_in_a_small_volume(oracle):
    """NaN / zero-area boxes make a frame irregular: its lists go through the general walk (walk_rest_kernel), the other frames'
    through the small one -- in one launch sequence, same results as the reference's loop"""
    rng = np.random.RandomState(31)
    F, B, C = 6, 200, 4
    boxes, scores = synth.video(4242, F, B, C)
    boxes[1, 5] = [np.nan, 3, 40, 50]
    boxes[1, 9, 2] = np.inf
    boxes[4, 0] = [30, 30, 29, 80]                 # zero width under the +1 convention
    boxes[4, 7] = [30, 30, 20, 80]                 # negative width
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3, None, cap=B)
    for (idx, cnt) in _nms_both(boxes, scores, 0.3):
        assert np.array_equal(cnt, wcnt) and np.array_equal(idx, widx)


def test_small_paths_random_sweep():
    """120 random small volumes (sizes around the 64 / 128 / 256 / 384 boundaries, dense / sparse, integer / fractional,
    tied scores, thresholds 0.05 .. 0.99, with and without a score threshold): the small kernels against the large-list ones"""
    rng = np.random.RandomState(2025)
    for it in range(120):
        B = int(rng.choice([1, 2, 3, 31, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 383, 384]))
        F, C = int(rng.randint(1, 6)), int(rng.randint(1, 7))
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
        st = None if rng.randint(2) else float(rng.uniform(0, 0.5))
        (i1, c1), (i0, c0) = _nms_both(boxes.astype(np.float32), scores.astype(np.float32), t, st)
        assert np.array_equal(c0, c1) and np.array_equal(i0, i1), (it, B, F, C, t, kind)
