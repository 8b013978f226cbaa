"""-m gpu: the bucketed per-(frame, class) lists (csrc/bucket_kernels.hpp, round 4) -- lists cut into score-ordered
buckets instead of sorted, alive entries ranked by the walk -- against the oracle (utils/nms.pyx:17-68 per list,
vdet/track.py:189-252 for the tracking kernels that read the lists' heads), on contexts created with VDET_BUCKETS=1
(the default is the LSD sort: the bucket path measured no faster) at more than 1024 boxes per frame, incl. the cases the entry values alone do not order (equal keys, keys of a
thin histogram bin that interpolate to the same rank) and the lists it hands back to the LSD sort."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _ctx(buckets="1"):
    """a context with the bucket path on (VDET_BUCKETS is read at vdet_create; the default is sorted lists)"""
    import os
    import torch
    from vdetlib_amd import _lib
    old = os.environ.get("VDET_BUCKETS")
    os.environ["VDET_BUCKETS"] = buckets
    try:
        return _lib.Context(torch.cuda.current_device())
    finally:
        if old is None:
            del os.environ["VDET_BUCKETS"]
        else:
            os.environ["VDET_BUCKETS"] = old


def _check_nms(oracle, boxes, scores, ctx, layout="FBC", score_thresh=None, expect_bucketed=True, max_fail=0):
    import torch
    from vdetlib_amd import ops
    F, B, C = scores.shape
    ts = torch.from_numpy(scores).cuda()
    if layout == "FCB":
        ts = ts.permute(0, 2, 1).contiguous()
    idx, cnt = ops.nms_volume(torch.from_numpy(boxes).cuda(), ts, 0.3, score_thresh=score_thresh, layout=layout, ctx=ctx)
    if expect_bucketed:
        assert ctx.query(10) == 1
        nfail = ctx.query(11)
        assert 0 <= nfail <= max_fail, nfail
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3, score_thresh=score_thresh, threads=8)
    assert np.array_equal(cnt.cpu().numpy(), wcnt)
    assert np.array_equal(idx.cpu().numpy(), widx)
    return int(ctx.query(11)) if expect_bucketed else -1


@pytest.mark.parametrize("kind,frac,B", [("perm", False, 2500), ("randn", False, 3000), ("perm", True, 1100), ("randn", False, 5000)])
def test_bucketed_nms_volume_vs_oracle(oracle, kind, frac, B):
    boxes, scores = synth.video(4100 + B, 3, B, 5, frac=frac, kind=kind)
    _check_nms(oracle, boxes, scores, _ctx())


def test_bucketed_nms_other_layout_and_threshold(oracle):
    """[F,C,B] float scores (the kernel's FLOATS form) and the score threshold (non-candidates leave the lists) in both layouts"""
    boxes, scores = synth.video(4201, 2, 2048, 4, kind="randn")
    ctx = _ctx()
    _check_nms(oracle, boxes, scores, ctx, layout="FCB")
    _check_nms(oracle, boxes, scores, ctx, layout="FCB", score_thresh=0.25)
    _check_nms(oracle, boxes, scores, ctx, layout="FBC", score_thresh=-0.5)
    _check_nms(oracle, boxes, scores, ctx, layout="FBC", score_thresh=10.0)       # nothing is a candidate


def _isolated(boxes, f, rows):
    """move the given boxes of frame f far away from everything (and from each other): they always survive"""
    for k, r in enumerate(rows):
        boxes[f, r] = np.array([5000 + 40 * k, 5000, 5000 + 40 * k + 20, 5020], np.float32)


def test_bucketed_ties_the_entry_values_do_not_decide(oracle):
    """(a) exact duplicates of a score (equal ord, the index decides), (b) ALIVE neighbours of a thin histogram bin whose
    keys differ by one ulp (equal ord, different keys: the walk ranks them by the full keys) at the low end of the lists,
    (c) the same at the top of a list, inside the head whose exact order bucket_kernel writes for the tracking kernels."""
    boxes, scores = synth.video(4301, 2, 2200, 3, kind="perm")
    rng = np.random.RandomState(5)
    # (a) duplicates, spread over the list
    for c in range(3):
        src = rng.permutation(2200)[:60]
        scores[0, src[:30], c] = scores[0, src[30:], c]
    # (b) 24 isolated boxes with scores one ulp apart in an almost empty octave
    rows = rng.permutation(2200)[:24]
    _isolated(boxes, 1, rows)
    tiny = np.float32(1e-20)
    vals = [tiny]
    for _ in range(23):
        vals.append(np.nextafter(vals[-1], np.float32(1), dtype=np.float32))
    scores[1, rows, 0] = np.array(vals, np.float32)[rng.permutation(24)]
    nfail = _check_nms(oracle, boxes, scores, _ctx(), max_fail=0)
    assert nfail == 0
    # (c) at the top: the head's exact order cannot come from the entry values alone
    boxes2, scores2 = synth.video(4302, 1, 1500, 2, kind="perm")
    rows = rng.permutation(1500)[:6]
    _isolated(boxes2, 0, rows)
    vals = [np.float32(3e20)]
    for _ in range(5):
        vals.append(np.nextafter(vals[-1], np.float32(np.inf), dtype=np.float32))
    scores2[0, rows, 1] = np.array(vals, np.float32)[rng.permutation(6)]
    import torch
    from vdetlib_amd import ops
    ctx = _ctx()
    tb, ts = torch.from_numpy(boxes2).cuda(), torch.from_numpy(scores2).cuda()
    ki, kc, tr, an, nt = ops.nms_track_volume(tb, ts, cap=1500, max_tracks=3, thres=0.0, ctx=ctx)
    assert ctx.query(10) == 1 and ctx.query(11) == 0            # (ties are settled by the full keys, not handed to the LSD kernel)
    widx, wcnt = oracle.nms_volume(boxes2, scores2, 0.3)
    assert np.array_equal(kc.cpu().numpy(), wcnt) and np.array_equal(ki.cpu().numpy(), widx)
    for c in range(2):
        wt, wa, wn = oracle.greedy_track_volume(boxes2, scores2[:, :, c], 0.3, 0.0, 3, 0.5, 0)
        assert nt[c] == wn and np.array_equal(an[c, :wn].cpu().numpy(), wa[:wn])
        assert np.array_equal(tr[c, :wn].cpu().numpy(), wt[:wn], equal_nan=True)


def test_quantised_scores_fall_back_to_the_lsd_sort(oracle):
    """heavily tied scores: a bucket would hold more than one wave's worth of keys -> every list is sorted by the LSD kernel"""
    boxes, scores = synth.video(4401, 2, 1300, 3, kind="perm")
    scores = (np.round(scores * 8) / 8).astype(np.float32)
    ctx = _ctx()
    nfail = _check_nms(oracle, boxes, scores, ctx, max_fail=6)
    assert nfail == 6


def test_irregular_frames_keep_sorted_lists(oracle):
    """a frame with a NaN coordinate is walked by the general kernels, which read whole sorted lists"""
    boxes, scores = synth.video(4501, 3, 1200, 2, kind="perm")
    boxes[1, 17, 2] = np.nan
    ctx = _ctx()
    nfail = _check_nms(oracle, boxes, scores, ctx, max_fail=2)
    assert nfail == 2


@pytest.mark.parametrize("cfg", [dict(seed=4601, F=6, B=1600, C=3, max_tracks=12, thres=0.0, max_frames=0, jitter=3),
                                 dict(seed=4602, F=5, B=2600, C=2, max_tracks=40, thres=0.3, max_frames=3, jitter=5)])
def test_bucketed_tracking_vs_oracle(oracle, cfg):
    """NMS survivors + greedy tubelets from bucketed lists: the tracking kernels read the exact heads bucket_kernel wrote
    and order further buckets on demand"""
    import torch
    from vdetlib_amd import ops
    boxes, scores = synth.coherent_video(cfg['seed'], cfg['F'], cfg['B'], cfg['C'], cfg['jitter'])
    ctx = _ctx()
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    ki, kc, tr, an, nt = ops.nms_track_volume(tb, ts, cap=cfg['B'], nms_thres=0.3, thres=cfg['thres'], max_tracks=cfg['max_tracks'],
                                              link_thres=0.5, max_frames=cfg['max_frames'], ctx=ctx)
    assert ctx.query(10) == 1 and ctx.query(11) == 0
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3, threads=8)
    assert np.array_equal(kc.cpu().numpy(), wcnt) and np.array_equal(ki.cpu().numpy(), widx)
    tr, an, nt = tr.cpu().numpy(), an.cpu().numpy(), nt.cpu().numpy()
    for c in range(cfg['C']):
        wt, wa, wn = oracle.greedy_track_volume(boxes, scores[:, :, c], 0.3, cfg['thres'], cfg['max_tracks'], 0.5, cfg['max_frames'])
        assert nt[c] == wn, (c, nt[c], wn)
        assert np.array_equal(an[c, :wn], wa[:wn]), c
        assert np.array_equal(tr[c, :wn], wt[:wn], equal_nan=True), c


def test_tracking_orders_every_bucket_on_demand(oracle):
    """lists cached by an NMS-only call carry no exact heads: the tracking call that reuses them (same boxes, same scores,
    cache on) materialises every list from its first bucket on (bucket_extend)"""
    import torch
    from vdetlib_amd import ops
    boxes, scores = synth.coherent_video(4701, 5, 1400, 2, 4)
    ctx = _ctx()
    ctx.set_cache(True)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    idx, cnt = ops.nms_volume(tb, ts, 0.3, ctx=ctx)
    assert ctx.query(10) == 1
    tr, an, nt = ops.track_volume(tb, ts, nms_thres=0.3, thres=0.0, max_tracks=25, link_thres=0.5, ctx=ctx)
    tr, an, nt = tr.cpu().numpy(), an.cpu().numpy(), nt.cpu().numpy()
    for c in range(2):
        wt, wa, wn = oracle.greedy_track_volume(boxes, scores[:, :, c], 0.3, 0.0, 25, 0.5, 0)
        assert nt[c] == wn, (c, nt[c], wn)
        assert np.array_equal(an[c, :wn], wa[:wn]), c
        assert np.array_equal(tr[c, :wn], wt[:wn], equal_nan=True), c


def test_bucket_path_and_lsd_path_agree_at_the_limit():
    """B = 16 384 (the entry format's index field) against the sorted lists of the same volume (the default path)"""
    import os
    import torch
    from vdetlib_amd import ops
    g = torch.Generator(device="cuda").manual_seed(48)
    F, B, C = 2, 16384, 3
    x1 = torch.rand(F, B, generator=g, device="cuda") * 3000
    y1 = torch.rand(F, B, generator=g, device="cuda") * 2000
    w = 10 + torch.rand(F, B, generator=g, device="cuda") * 290
    h = 10 + torch.rand(F, B, generator=g, device="cuda") * 290
    boxes = torch.stack([x1, y1, x1 + w, y1 + h], -1).round().contiguous()
    scores = torch.randn(F, B, C, generator=g, device="cuda")
    a = _ctx()
    ia, ca = ops.nms_volume(boxes, scores, 0.3, ctx=a)
    assert a.query(10) == 1 and a.query(11) == 0
    b = _ctx("0")                                            # sorted lists (the default, whatever the suite's environment says)
    ib, cb = ops.nms_volume(boxes, scores, 0.3, ctx=b)
    assert b.query(10) == 0
    assert torch.equal(ca, cb) and torch.equal(ia, ib)
