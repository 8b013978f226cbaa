"""-m gpu: the adjacency build of regular large frames -- the DIRECT lists (iou_bits_sym_kernel<true, true>: entries straight
from the predicate blocks into fixed slots per row + adj_finish_kernel; the default), the rows kernel behind the bit matrix
(csrc/adjrows_kernels.hpp: one workgroup per 64-row strip, slab offsets from a pre-pass; VDET_DIRECT_LISTS=0, and what a frame
with a row of more neighbours than a slot holds falls back to) -- on the shapes their paths branch on, against the oracle's
per-(frame, class) NMS (utils/nms.pyx:17-68) and against adj_build_kernel (VDET_ADJ_ROWS=0):
  * a strip whose slab does not fit the 16 KB LDS stage (a dense frame: entries go straight to the pool),
  * more than 64 existing column words per strip (wide boxes in a narrow frame, low threshold: the extra round trips),
  * B not a multiple of 64 / of 256, B just above the small-frame limit (385), B = 17 400 (272 word-rows > 256 threads),
  * frames of different kinds in ONE volume next to an irregular frame (zero-area box: adj_build_kernel's rows),
  * re-scoring candidates from the lists (order inside a list differs between the two kernels: the arg-max must not)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _boxes(rng, B, frame_w, frame_h, wmax, hmax):
    x1 = rng.randint(0, frame_w - 8, B)
    y1 = rng.randint(0, frame_h - 8, B)
    w = rng.randint(6, wmax, B)
    h = rng.randint(6, hmax, B)
    return np.stack([x1, y1, np.minimum(x1 + w, frame_w - 1), np.minimum(y1 + h, frame_h - 1)], 1).astype(np.float32)


def _nms_both_kernels(monkeypatch, boxes, scores, thresh, cap):
    import torch
    from vdetlib_amd import ops, _lib
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    out = []
    for rows, direct in (("1", "1"), ("0", "1"), ("1", "0")):       # direct lists (default) / lane per row / rows kernel
        monkeypatch.setenv("VDET_ADJ_ROWS", rows)
        monkeypatch.setenv("VDET_DIRECT_LISTS", direct)
        cx = _lib.Context(torch.cuda.current_device())
        idx, cnt = ops.nms_volume(tb, ts, thresh, cap=cap, ctx=cx)
        out.append((idx.cpu().numpy(), cnt.cpu().numpy(), cx.query(2)))
        cx.close()
    return out


CASES = {
    # name: (F, B, C, frame_w, frame_h, wmax, hmax, thresh)
    "dense_unstaged": (2, 1500, 3, 260, 200, 200, 160, 0.3),         # degree ~ 500: 64 lists > 8 192 entries
    "many_words": (2, 6000, 2, 640, 720, 300, 300, 0.05),           # window ~ 90 % of the frame: > 64 existing words of 94
    "ragged_385": (3, 385, 4, 1280, 720, 300, 300, 0.3),
    "ragged_777": (3, 777, 4, 1280, 720, 300, 300, 0.3),
    "ragged_4100": (2, 4100, 2, 1280, 720, 300, 300, 0.5),
    "wide_17400": (1, 17400, 1, 1280, 720, 200, 200, 0.5),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_adjacency_rows_against_oracle_and_lane_per_row_kernel(name, oracle, monkeypatch):
    F, B, C, fw, fh, wmax, hmax, t = CASES[name]
    rng = np.random.RandomState(len(name) * 131 + B)
    boxes = np.stack([_boxes(rng, B, fw, fh, wmax, hmax) for _ in range(F)])
    scores = synth.tie_free_scores(rng, F * B * C).reshape(F, B, C).astype(np.float32)
    (idx, cnt, reg), (idx0, cnt0, reg0), (idx1, cnt1, reg1) = _nms_both_kernels(monkeypatch, boxes, scores, t, cap=B)
    assert reg == 1 and reg0 == 1 and reg1 == 1         # every frame regular: the direct lists / the rows kernel ran
    assert np.array_equal(cnt, cnt0) and np.array_equal(idx, idx0)
    assert np.array_equal(cnt, cnt1) and np.array_equal(idx, idx1)
    widx, wcnt = oracle.nms_volume(boxes, scores, t, cap=B)
    assert np.array_equal(cnt, wcnt) and np.array_equal(idx, widx)


def test_adjacency_rows_next_to_an_irregular_frame(oracle, monkeypatch):
    """frame 1 holds a zero-area box (irregular: general predicate kernel + adj_build_kernel with zero-union tags), frames 0 and 2
    take the rows kernel -- in one launch; the reference's ZeroDivisionError semantics stay those of the oracle"""
    rng = np.random.RandomState(99)
    F, B, C = 3, 900, 3
    boxes = np.stack([_boxes(rng, B, 1280, 720, 300, 300) for _ in range(F)])
    boxes[1, 17] = [100, 100, 99, 250]                  # zero width under the +1 convention, far from everything that matters?
    scores = synth.tie_free_scores(rng, F * B * C).reshape(F, B, C).astype(np.float32)
    import torch
    from vdetlib_amd import ops
    try:
        widx, wcnt = oracle.nms_volume(boxes, scores, 0.3, cap=B)
    except ZeroDivisionError:
        with pytest.raises(ZeroDivisionError):
            ops.nms_volume(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.3, cap=B)
        return
    idx, cnt = ops.nms_volume(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.3, cap=B)
    assert np.array_equal(cnt.cpu().numpy(), wcnt) and np.array_equal(idx.cpu().numpy(), widx)


def test_rescoring_candidates_do_not_depend_on_list_order(monkeypatch):
    """tubelets + spatial max-pool re-scoring (candidates = graph neighbours, csrc/track_kernels.hpp rescore_adj_one) with the
    lists of either kernel: identical outputs"""
    import torch
    from vdetlib_amd import ops, _lib
    boxes, scores = synth.video(4242, 12, 2000, 6)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    res = []
    for rows, direct in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("VDET_ADJ_ROWS", rows)
        monkeypatch.setenv("VDET_DIRECT_LISTS", direct)
        cx = _lib.Context(torch.cuda.current_device())
        cx.set_cache(True)
        ki, kc, tr, an, nt = ops.nms_track_volume(tb, ts, nms_thres=0.3, thres=0.5, max_tracks=5, link_thres=0.5, cap=2000, ctx=cx)
        det, tp, tbx = ops.rescore_tracks(tr, nt, tb, ts, overlap_thres=0.7, window=3, ctx=cx)
        res.append([x.cpu().numpy() for x in (ki, kc, tr, an, nt, det, tp, tbx)])
        cx.close()
    for a, b, d in zip(*res):
        assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a, d, equal_nan=True)


@pytest.mark.parametrize("slot", [8, 96, 384])
def test_direct_lists_slot_sizes_and_the_fallback(slot, oracle, monkeypatch):
    """VDET_DIRECT_CAP = entries per row slot.  8 / 96: rows with more neighbours than a slot holds (degree ~ 90 here) -- the build
    latches the overflow, the context takes the bit-matrix path for good and the results are the oracle's; 384: (nearly) everything fits.
    Synchronous and asynchronous (the overflow of an asynchronous build is a RetryError at sync, the repeat succeeds)."""
    import torch
    from vdetlib_amd import ops, _lib
    rng = np.random.RandomState(7 + slot)
    F, B, C = 3, 2500, 3
    boxes = np.stack([_boxes(rng, B, 640, 480, 300, 300) for _ in range(F)])
    scores = synth.tie_free_scores(rng, F * B * C).reshape(F, B, C).astype(np.float32)
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3, cap=B)
    monkeypatch.setenv("VDET_DIRECT_CAP", str(slot))
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    cx = _lib.Context(torch.cuda.current_device())
    idx, cnt = ops.nms_volume(tb, ts, 0.3, cap=B, ctx=cx)
    assert np.array_equal(cnt.cpu().numpy(), wcnt) and np.array_equal(idx.cpu().numpy(), widx)
    cx.close()
    # asynchronous: a sparse video first (its direct build fits a slot of 96 or 384 entries), then the dense one
    cx = _lib.Context(torch.cuda.current_device())
    cx.set_cache(True)
    cx.set_async(True)
    sparse = np.stack([_boxes(rng, B, 4000, 3000, 120, 120) for _ in range(F)])
    ops.nms_volume(torch.from_numpy(sparse).cuda(), ts, 0.3, cap=B, ctx=cx)
    cx.invalidate()
    idx, cnt = ops.nms_volume(tb, ts, 0.3, cap=B, ctx=cx)      # sync=True: repeats the enqueue by itself after a RetryError
    assert np.array_equal(cnt.cpu().numpy(), wcnt) and np.array_equal(idx.cpu().numpy(), widx)
    cx.close()


def test_direct_lists_on_groups_of_mixed_sizes(oracle):
    """vid_nms (utils/nms.pyx:71-125) through the drop-in module: ONE plan whose frames hold 1 / 3 / 64 / 385 / 500 / 1 300 detections
    (the largest takes the plan to the direct lists; the small groups and the singleton ride along: partial tiles, tiles with one
    valid quartile, a group without a graph), fractional and integer boxes, against the oracle's kept list."""
    from vdetlib_amd.utils import cython_nms as cnms
    rng = np.random.RandomState(20)
    for integer in (True, False):
        rows = []
        for frame, n in enumerate([1, 3, 64, 385, 500, 1300, 2]):
            b = _boxes(rng, n, 900, 700, 260, 260)
            if not integer:
                b = b + rng.rand(n, 4).astype(np.float32) * 0.5
            rows.append(np.hstack([np.full((n, 1), frame + 1, np.float32), b, rng.rand(n, 1).astype(np.float32)]))
        d = np.vstack(rows).astype(np.float32)
        d = d[rng.permutation(len(d))]
        assert cnms.vid_nms(d, 0.3) == oracle.vid_nms(d, 0.3)
