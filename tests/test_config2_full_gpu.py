"""-m gpu: BASELINE configs[1] + [2] at FULL size -- one video of 300 frames x 10 000 boxes x 200 classes
through the benchmarked calls (vdet_volume_pass, vdet_nms_track_volume, vdet_rescore_tracks) -- checked
against the oracle on samples that cover every structural boundary of the device path: the four
bit-matrix batches (32 frames incl. 0 / 84 / 85 / 170 / 299 x ALL 200 classes of NMS survivors, all host threads),
box tiles at both ends and in the middle of the volume pass (12 % of the boxes x all 300 frames x all classes of both
temporal outputs), and ALL TEN tubelets + their re-scoring of six classes -- among them, when the video has one, a
class whose anchors the memo warm-up did not predict -- with the oracle's per-class pipeline on one host process each."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F, B, C = 300, 10000, 200
TAPS = [0.25, 0.5, 0.25]


@pytest.fixture(scope="module")
def full_run():
    import torch
    from vdetlib_amd import ops, _lib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda", torch.cuda.current_device())
    boxes, scores = bench.synth_video_cuda(torch, 31337, F, B, C, dev)
    cx = _lib.Context(dev.index)
    cx.set_cache(True)
    pooled, conv = ops.volume_pass(scores, 3, TAPS, ctx=cx)
    keep_idx, keep_cnt, tracks, anchors, ntracks = ops.nms_track_volume(
        boxes, scores, nms_thres=0.3, thres=0.9, max_tracks=10, link_thres=0.5, cap=2048, ctx=cx)
    assert cx.query(2) == 1                      # every frame regular: the K1s / lazy-list fast paths ran
    det, tpool, tboxes = ops.rescore_tracks(tracks, ntracks, boxes, scores, overlap_thres=0.7, window=3, ctx=cx)
    torch.cuda.synchronize()
    out = dict(boxes=boxes, scores=scores, pooled=pooled, conv=conv, keep_idx=keep_idx, keep_cnt=keep_cnt, tracks=tracks,
               anchors=anchors, ntracks=ntracks, det=det, tpool=tpool, tboxes=tboxes)
    yield out
    cx.close()


def test_full_volume_nms_survivors_sampled_frames(full_run, oracle):
    r = full_run
    # first / last frame, both sides of the bit-matrix batch borders, and every tenth frame: 32 frames x 200 classes
    frames = sorted(set([0, 84, 85, 170, 299] + list(range(5, F, 11))))
    assert len(frames) >= 30
    hb = r['boxes'][frames].cpu().numpy()
    hs = r['scores'][frames].contiguous().cpu().numpy()
    nthr = max(1, len(os.sched_getaffinity(0)))
    widx, wcnt = oracle.nms_volume(hb, hs, 0.3, cap=2048, threads=nthr)
    gcnt = r['keep_cnt'][frames].cpu().numpy()
    gidx = r['keep_idx'][frames].cpu().numpy()
    assert np.array_equal(gcnt, wcnt)
    assert np.array_equal(gidx, widx)
    assert 1000 < gcnt.mean() < 2048


def test_full_volume_temporal_outputs_sampled_tiles(full_run, oracle):
    r = full_run
    for b0, b1 in ((0, 400), (4800, 5210), (B - 400, B)):        # 1 210 boxes (12 %): dozens of 32-box tiles incl. the ragged last one
        hs = r['scores'][:, b0:b1].contiguous().cpu().numpy()
        assert np.array_equal(r['pooled'][:, b0:b1].cpu().numpy(), oracle.temporal_maxpool(hs, 3))
        # (the volume pass and the oracle do the same f32 operations in the same order)
        np.testing.assert_allclose(r['conv'][:, b0:b1].cpu().numpy(), oracle.temporal_conv(hs, TAPS, 0.0, 0.0), rtol=0, atol=1e-6)


def _classes_with_unpredicted_anchors(r, m=16):
    """The memo warm-up predicts, per class, the best m detections by (score desc, flat index asc) among the first two
    entries of every frame's sorted list (track_warm_anchors_kernel).  An anchor of the final result that is not among
    them was mispredicted: its tubelet came from the tracking loop's own link, not from the materialised chains."""
    import torch
    sc = r['scores']                                               # [F,B,C]
    top = torch.topk(sc, 2, dim=1)                                 # values / indices [F,2,C]
    flat = top.indices + (torch.arange(F, device=sc.device) * B)[:, None, None]
    v = top.values.permute(2, 0, 1).reshape(C, -1)                 # [C, 2F]
    fl = flat.permute(2, 0, 1).reshape(C, -1)
    # descending score, ties by ascending flat index: sort by flat first (stable), then by score
    o1 = torch.argsort(fl, dim=1, stable=True)
    v, fl = torch.gather(v, 1, o1), torch.gather(fl, 1, o1)
    o2 = torch.argsort(v, dim=1, descending=True, stable=True)
    pred = torch.gather(fl, 1, o2)[:, :m]                          # [C, m]
    an = r['anchors']                                              # [C,T,3]: 1-based frame, box, score
    aflat = ((an[:, :, 0] - 1) * B + an[:, :, 1]).long()           # [C,T]
    hit = (aflat[:, :, None] == pred[:, None, :]).any(dim=2)       # [C,T]
    return [int(c) for c in torch.nonzero(~hit.all(dim=1)).flatten().tolist()]


def test_full_volume_all_tubelets_six_classes(full_run, tmp_path):
    """vdet/track.py:189-252 + vdet/tubelet_cls.py:493-535, :284-303, :386-414 at full size: ALL ten tubelets of six
    classes -- anchors, rows, spatial max-pool / regressed boxes, completion, temporal max-pool -- against
    oracle.rescored_tubelets (lists pruned several times, warm-anchor mispredictions, materialised-chain copies:
    everything the first two tracks never exercise)."""
    import oracle_pool
    r = full_run
    nt_dev = r['ntracks'].cpu().numpy()
    assert (nt_dev == 10).all()                  # 3 M U(0,1) scores per class: ten anchors above 0.9 always exist
    T = 10
    missed = _classes_with_unpredicted_anchors(r)
    classes = (missed[:2] + [c for c in (0, 41, 99, 137, 163, 199) if c not in missed[:2]])[:6]
    hb = r['boxes'].cpu().numpy()
    cols = {c: r['scores'][:, :, c].contiguous().cpu().numpy() for c in classes}
    opts = dict(nms_thres=0.3, thres=0.9, max_tracks=T, link_thres=0.5, pool_thres=0.7, window=3)
    want = oracle_pool.rescored_tubelets_per_class(hb, cols, opts, tmp_path)
    for c in classes:
        wt, wn, wpool, wbx, wdet = want[c]
        assert wn == T, (c, wn)
        gt = r['tracks'][c].cpu().numpy()
        assert np.array_equal(gt, wt, equal_nan=True), c
        # anchors: 1-based frame, box index, score -- the row with link score 1 of every tubelet is its anchor's frame
        ga = r['anchors'][c].cpu().numpy()
        for t in range(T):
            f0 = int(ga[t, 0]) - 1
            assert wt[t, f0, 4] == 1.0 and np.array_equal(np.trunc(hb[f0, int(ga[t, 1])]), wt[t, f0, :4])
            assert ga[t, 2] == cols[c][f0, int(ga[t, 1])]
        has = ~np.isnan(wt[:, :, 0])
        gd, gp, gb = r['det'][c].cpu().numpy(), r['tpool'][c].cpu().numpy(), r['tboxes'][c].cpu().numpy()
        assert np.array_equal(np.isnan(gp), ~has) and np.array_equal(np.isnan(gd), ~has)
        np.testing.assert_allclose(gd[has], wdet[has], rtol=0, atol=1e-9)
        np.testing.assert_allclose(gp[has], wpool[has], rtol=0, atol=1e-9)
        assert np.array_equal(gb[has], wbx[has])


def test_full_volume_size_independent_properties(full_run):
    """Whole-volume checks that need no oracle: survivor lists are score-descending everywhere, counts are
    consistent with the padding, every track row is a proposal of its frame, temporal max-pool dominates its input."""
    import torch
    r = full_run
    cnt, idx = r['keep_cnt'], r['keep_idx']
    assert int(cnt.min()) > 0 and int(cnt.max()) <= 2048
    ar = torch.arange(idx.shape[2], device=idx.device)[None, None, :]
    valid = ar < cnt[:, :, None]
    assert bool(((idx >= 0) == valid).all()) and int(idx.max()) < B
    for f in (3, 150, 298):                       # descending scores along every survivor list of the frame
        sc = r['scores'][f].t().contiguous()                                     # [C,B]
        g = torch.gather(sc, 1, idx[f].clamp(min=0).long())
        g = torch.where(valid[f], g, torch.full_like(g, -1.0))
        assert bool((g[:, 1:] <= g[:, :-1]).all())
    assert bool((r['pooled'] >= r['scores']).all())
    assert bool((r['pooled'][1:-1] == torch.maximum(torch.maximum(r['scores'][:-2], r['scores'][1:-1]), r['scores'][2:])).all())
    tr = r['tracks']
    has = ~torch.isnan(tr[..., 0])
    assert bool(has.any(dim=2).all())
    # a tracked box (other than the int-truncated anchor) is an int-truncated proposal of its frame
    c, t = 5, 3
    for f in (0, 100, 299):
        if bool(has[c, t, f]):
            row = tr[c, t, f, :4]
            assert bool((torch.trunc(r['boxes'][f]) == row[None]).all(dim=1).any())


def test_full_volume_fast_paths_equal_the_plain_ones(full_run, monkeypatch):
    """All 200 classes x 10 tracks of the full video: the link memo + warm-up and the graph-neighbour re-scoring (the
    benchmarked paths) against the round-1 kernels they replace (every link step scans its window, every tubelet box
    scans its window), which the oracle tests above and the golden tests pin."""
    import torch
    from vdetlib_amd import ops, _lib
    r = full_run
    monkeypatch.setenv("VDET_LINK_MEMO", "0")
    monkeypatch.setenv("VDET_RESCORE_ADJ", "0")
    monkeypatch.setenv("VDET_WALK_CAREFUL", "1")
    monkeypatch.setenv("VDET_TRACK_LOOP", "0")
    cx = _lib.Context(torch.cuda.current_device())
    cx.set_cache(True)
    keep_idx, keep_cnt, tracks, anchors, ntracks = ops.nms_track_volume(
        r['boxes'], r['scores'], nms_thres=0.3, thres=0.9, max_tracks=10, link_thres=0.5, cap=2048, ctx=cx)
    det, tpool, tboxes = ops.rescore_tracks(tracks, ntracks, r['boxes'], r['scores'], overlap_thres=0.7, window=3, ctx=cx)
    for name, a, b in (("keep_cnt", keep_cnt, r['keep_cnt']), ("keep_idx", keep_idx, r['keep_idx']), ("ntracks", ntracks, r['ntracks']),
                       ("anchors", anchors, r['anchors']), ("tracks", tracks, r['tracks']), ("det", det, r['det']),
                       ("tpool", tpool, r['tpool']), ("tboxes", tboxes, r['tboxes'])):
        assert torch.equal(a.nan_to_num(-7.0), b.nan_to_num(-7.0)), name
    cx.close()


def _variant_video(torch, kind, dev):
    """the c2 volume with other inputs than the benchmarked ones: fractional boxes (K1s's quotient fallback band, the int
    truncation of tracked boxes), N(0,1) scores (both signs, many exponents: other histogram bins / radix digits), a
    COHERENT video (every proposal persists with +-3 px of jitter and keeps 80 % of its score: distinct tubelets per class)"""
    g = torch.Generator(device=dev).manual_seed({"frac": 411, "randn": 412, "coherent": 413}[kind])
    if kind == "coherent":
        base = torch.rand(B, 4, generator=g, device=dev)
        x1, y1 = base[:, 0] * 1230, base[:, 1] * 670
        bb = torch.stack([x1, y1, torch.clamp(x1 + 10 + base[:, 2] * 290, max=1279), torch.clamp(y1 + 10 + base[:, 3] * 290, max=719)], -1)
        boxes = (bb[None] + torch.randint(-3, 4, (F, B, 4), generator=g, device=dev)).round()
        boxes[..., 2:] = torch.maximum(boxes[..., 2:], boxes[..., :2] + 4)
        scores = torch.rand(F, B, C, generator=g, device=dev).mul_(0.2).add_(0.8 * torch.rand(B, C, generator=g, device=dev)[None])
        return boxes.contiguous(), scores.contiguous(), 0.9
    x1 = torch.rand(F, B, generator=g, device=dev) * 1230
    y1 = torch.rand(F, B, generator=g, device=dev) * 670
    w = 10 + torch.rand(F, B, generator=g, device=dev) * 290
    h = 10 + torch.rand(F, B, generator=g, device=dev) * 290
    boxes = torch.stack([x1, y1, torch.clamp(x1 + w, max=1279), torch.clamp(y1 + h, max=719)], -1)
    boxes = boxes.contiguous() if kind == "frac" else boxes.round().contiguous()
    if kind == "randn":
        return boxes, torch.randn(F, B, C, generator=g, device=dev), 2.5
    return boxes, torch.rand(F, B, C, generator=g, device=dev), 0.9


@pytest.mark.parametrize("kind", ["frac", "randn", "coherent"])
def test_full_volume_other_inputs(kind, oracle, tmp_path):
    """at FULL size the oracle had only met U(0,1) scores on integer boxes.  Per variant: the NMS
    survivors of 9 frames x all 200 classes, and all ten tubelets + re-scoring of two classes."""
    import torch
    import oracle_pool
    from vdetlib_amd import ops, _lib
    dev = torch.device("cuda", torch.cuda.current_device())
    boxes, scores, thres = _variant_video(torch, kind, dev)
    cx = _lib.Context(dev.index)
    cx.set_cache(True)
    keep_idx, keep_cnt, tracks, anchors, ntracks = ops.nms_track_volume(
        boxes, scores, nms_thres=0.3, thres=thres, max_tracks=10, link_thres=0.5, cap=2048, ctx=cx)
    det, tpool, tboxes = ops.rescore_tracks(tracks, ntracks, boxes, scores, overlap_thres=0.7, window=3, ctx=cx)
    frames = [0, 1, 84, 85, 149, 170, 171, 298, 299]
    hb = boxes[frames].cpu().numpy()
    hs = scores[frames].contiguous().cpu().numpy()
    widx, wcnt = oracle.nms_volume(hb, hs, 0.3, cap=2048, threads=max(1, len(os.sched_getaffinity(0))))
    assert np.array_equal(keep_cnt[frames].cpu().numpy(), wcnt)
    assert np.array_equal(keep_idx[frames].cpu().numpy(), widx)
    classes = [7, 150]
    hball = boxes.cpu().numpy()
    cols = {c: scores[:, :, c].contiguous().cpu().numpy() for c in classes}
    want = oracle_pool.rescored_tubelets_per_class(hball, cols, dict(nms_thres=0.3, thres=thres, max_tracks=10, link_thres=0.5,
                                                                     pool_thres=0.7, window=3), tmp_path)
    for c in classes:
        wt, wn, wpool, wbx, wdet = want[c]
        assert int(ntracks[c]) == wn, (kind, c, int(ntracks[c]), wn)
        assert np.array_equal(tracks[c, :wn].cpu().numpy(), wt[:wn], equal_nan=True), (kind, c)
        has = ~np.isnan(wt[:wn, :, 0])
        gd, gp, gb = det[c, :wn].cpu().numpy(), tpool[c, :wn].cpu().numpy(), tboxes[c, :wn].cpu().numpy()
        np.testing.assert_allclose(gd[has], wdet[:wn][has], rtol=0, atol=1e-9)
        np.testing.assert_allclose(gp[has], wpool[:wn][has], rtol=0, atol=1e-9)
        assert np.array_equal(gb[has], wbx[:wn][has])
    cx.close()
