"""-m gpu: BASELINE configs[1] + [2] at FULL size -- one video of 300 frames x 10 000 boxes x 200 classes
through the benchmarked calls (vdet_volume_pass, vdet_nms_track_volume, vdet_rescore_tracks) -- checked
against the oracle on samples that cover every structural boundary of the device path: the four
bit-matrix batches (frames 0 / 84 / 85 / 170 / 299 x ALL 200 classes of NMS survivors), box tiles at both
ends and in the middle of the volume pass (all 300 frames x all classes of both temporal outputs), and the
first tubelets (+ re-scoring) of two classes."""
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
    frames = [0, 84, 85, 170, 299]               # first / last frame and both sides of the bit-matrix batch borders
    hb = r['boxes'][frames].cpu().numpy()
    hs = r['scores'][frames].contiguous().cpu().numpy()
    nthr = max(1, min(64, len(os.sched_getaffinity(0))))
    widx, wcnt = oracle.nms_volume(hb, hs, 0.3, cap=2048, threads=nthr)
    gcnt = r['keep_cnt'][frames].cpu().numpy()
    gidx = r['keep_idx'][frames].cpu().numpy()
    assert np.array_equal(gcnt, wcnt)
    assert np.array_equal(gidx, widx)
    assert 1000 < gcnt.mean() < 2048


def test_full_volume_temporal_outputs_sampled_tiles(full_run, oracle):
    r = full_run
    for b0, b1 in ((0, 40), (4980, 5030), (B - 48, B)):          # spans several 32-box tiles incl. the ragged last one
        hs = r['scores'][:, b0:b1].contiguous().cpu().numpy()
        assert np.array_equal(r['pooled'][:, b0:b1].cpu().numpy(), oracle.temporal_maxpool(hs, 3))
        # (the volume pass and the oracle do the same f32 operations in the same order)
        np.testing.assert_allclose(r['conv'][:, b0:b1].cpu().numpy(), oracle.temporal_conv(hs, TAPS, 0.0, 0.0), rtol=0, atol=1e-6)


def test_full_volume_tubelets_two_classes(full_run, oracle):
    r = full_run
    hb = r['boxes'].cpu().numpy()
    nt_dev = r['ntracks'].cpu().numpy()
    assert (nt_dev == 10).all()                  # 3 M U(0,1) scores per class: ten anchors above 0.9 always exist
    T = 2                                        # greedy prefix: the first T tracks do not depend on max_tracks
    for c in (0, 137):
        hs = r['scores'][:, :, c].contiguous().cpu().numpy()
        wt, wa, wn = oracle.greedy_track_volume(hb, hs, 0.3, 0.9, T, 0.5, 0)
        assert wn == T
        assert np.array_equal(r['anchors'][c, :T].cpu().numpy(), wa[:T])
        assert np.array_equal(r['tracks'][c, :T].cpu().numpy(), wt[:T], equal_nan=True)
        # re-scoring of those tubelets: spatial max-pool (f64 IoU > 0.7), completion, temporal max-pool
        gd = r['det'][c, :T].cpu().numpy()
        gp = r['tpool'][c, :T].cpu().numpy()
        gb = r['tboxes'][c, :T].cpu().numpy()
        for t in range(T):
            fr = [f for f in range(F) if not np.isnan(wt[t, f, 0])]
            s, bx = [], []
            for f in fr:
                ss, bb, _ = oracle.spatial_maxpool([wt[t, f, :4]], hb[f], hs[f], 0.7)
                s.append(ss[0]); bx.append(bb[0])
            comp = oracle.score_completion(s)
            pool = [max(comp[g] if 0 <= g < len(comp) else -1e5 for g in (i - 1, i, i + 1)) for i in range(len(comp))]
            np.testing.assert_allclose(gd[t, fr], comp, rtol=0, atol=1e-9)
            np.testing.assert_allclose(gp[t, fr], pool, rtol=0, atol=1e-9)
            assert np.array_equal(gb[t, fr], np.asarray(bx, np.float32))
            assert np.isnan(gp[t]).sum() == F - len(fr)


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
