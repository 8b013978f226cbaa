"""vdet_video_batch / vdet_volume_pass_batch: V small videos concatenated along F (BASELINE configs[0] / [4] shapes) give,
video by video, exactly what the single-video entry points give -- NMS survivors, tubelets, anchors, re-scored tubelets,
temporal outputs -- and a sample of them is checked against the oracle directly."""
import numpy as np
import pytest
import torch

import synth
from vdetlib_amd import ops, _lib

pytestmark = pytest.mark.gpu

KW = dict(nms_thres=0.3, thres=0.2, max_tracks=4, link_thres=0.4)


def _videos(frames, B, C, seed=300, irregular_at=None):
    vids = []
    for v, f in enumerate(frames):
        b, s = synth.coherent_video(seed + v, f, B, C)
        if irregular_at is not None and v == irregular_at[0]:
            b[irregular_at[1], 3] = [30, 30, 29, 40]          # zero width: that frame is irregular (eager track_det_nms)
            s[irregular_at[1], 3] = 0.01
        vids.append((b, s))
    off = np.concatenate([[0], np.cumsum(frames)])
    boxes = torch.from_numpy(np.concatenate([b for b, _ in vids])).cuda()
    scores = torch.from_numpy(np.concatenate([s for _, s in vids])).cuda()
    return vids, off, boxes, scores


def _eq(a, b):
    return torch.equal(a.nan_to_num(-7.0), b.nan_to_num(-7.0))


@pytest.mark.parametrize("irregular,B", [(None, 150), ((2, 5), 150), (None, 151)])
def test_batch_equals_one_video_at_a_time(irregular, B):
    frames = [7, 1, 12, 3, 9]
    C = 6                       # (B odd: the compact u16 index pads every second frame)
    vids, off, boxes, scores = _videos(frames, B, C, irregular_at=irregular)
    cx = _lib.Context(torch.cuda.current_device())
    out = ops.video_batch(boxes, scores, off, overlap_thres=0.6, window=3, ctx=cx, **KW)
    one = _lib.Context(torch.cuda.current_device())
    one.set_cache(True)
    for v, (b, s) in enumerate(vids):
        tb, ts = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
        ki, kc, tr, an, nt = ops.nms_track_volume(tb, ts, ctx=one, **KW)
        det, pool, bx = ops.rescore_tracks(tr, nt, tb, ts, overlap_thres=0.6, window=3, ctx=one)
        f0, f1 = int(off[v]), int(off[v + 1])
        assert torch.equal(out["keep_cnt"][f0:f1], kc) and torch.equal(out["keep_idx"][f0:f1], ki), v
        assert torch.equal(out["ntracks"][v], nt), v
        for c in range(C):
            n = int(nt[c])
            assert _eq(out["anchors"][v, c, :n], an[c, :n]), (v, c)
            assert _eq(out["tracks"][v][c, :n], tr[c, :n]), (v, c)
            assert _eq(out["det"][v][c, :n], det[c, :n]) and _eq(out["pooled"][v][c, :n], pool[c, :n]), (v, c)
            assert _eq(out["tboxes"][v][c, :n], bx[c, :n]), (v, c)
    cx.close(); one.close()


def test_batch_sample_against_the_oracle(oracle):
    frames = [6, 10, 4]
    B, C = 120, 3
    vids, off, boxes, scores = _videos(frames, B, C, seed=900)
    out = ops.video_batch(boxes, scores, off, overlap_thres=0.6, window=3, **KW)
    v = 1
    b, s = vids[v]
    wt, wn, wpool, wbx = oracle.rescored_tubelets(b, s, KW["nms_thres"], KW["thres"], KW["max_tracks"], KW["link_thres"], 0.6, 3)
    assert np.array_equal(out["ntracks"][v].cpu().numpy(), wn)
    for c in range(C):
        n = int(wn[c])
        assert np.array_equal(out["tracks"][v][c, :n].cpu().numpy(), wt[c, :n], equal_nan=True)
        np.testing.assert_allclose(out["pooled"][v][c, :n].cpu().numpy(), wpool[c, :n], rtol=0, atol=1e-9)
        assert np.array_equal(out["tboxes"][v][c, :n].cpu().numpy(), wbx[c, :n], equal_nan=True)
    widx, wcnt = oracle.nms_volume(b, s, KW["nms_thres"])
    f0, f1 = int(off[v]), int(off[v + 1])
    assert np.array_equal(out["keep_cnt"][f0:f1].cpu().numpy(), wcnt) and np.array_equal(out["keep_idx"][f0:f1].cpu().numpy(), widx)


@pytest.mark.parametrize("C,window", [(8, 3), (8, 5), (6, 3), (5, 7)])
def test_volume_pass_stops_at_video_borders(C, window):
    """fast tiled kernel (C % 4 == 0, window 3 / 5), the two-operator vec4 kernel and the scalar fallback"""
    frames = [5, 1, 2, 9, 3]
    B = 37 if C == 5 else 64
    rng = np.random.default_rng(C * 10 + window)
    parts = [rng.standard_normal((f, B, C)).astype(np.float32) for f in frames]
    off = np.concatenate([[0], np.cumsum(frames)])
    taps = list(rng.standard_normal(window).astype(np.float32))
    vol = torch.from_numpy(np.concatenate(parts)).cuda()
    pooled, conv = ops.volume_pass(vol, window, taps, frame_off=off)
    for v, p in enumerate(parts):
        t = torch.from_numpy(p).cuda()
        pm, pc = ops.volume_pass(t, window, taps)
        f0, f1 = int(off[v]), int(off[v + 1])
        assert torch.equal(pooled[f0:f1], pm) and torch.equal(conv[f0:f1], pc), v


def test_batch_options_and_limits():
    """no tracks / no NMS output / no re-scoring; one video == the single-video call; offsets are validated"""
    frames = [4, 6]
    vids, off, boxes, scores = _videos(frames, 90, 3, seed=40)
    out = ops.video_batch(boxes, scores, off, max_tracks=0, nms=True, rescore=False, nms_thres=0.3)
    i0, c0 = ops.nms_volume(boxes, scores, 0.3)
    assert torch.equal(out["keep_cnt"], c0) and torch.equal(out["keep_idx"], i0) and int(out["ntracks"].sum()) == 0
    out = ops.video_batch(boxes, scores, off, nms=False, rescore=False, **KW)
    assert out["keep_cnt"] is None and out["pooled"] == [] and len(out["tracks"]) == 2
    b, s = vids[1]
    tb, ts = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    tr, an, nt = ops.track_volume(tb, ts, **KW)
    assert torch.equal(out["ntracks"][1], nt) and _eq(out["tracks"][1], tr)
    one = ops.video_batch(tb, ts, [0, 6], nms=False, rescore=False, **KW)
    assert torch.equal(one["ntracks"][0], nt) and _eq(one["tracks"][0], tr)
    with pytest.raises(ValueError):
        ops.video_batch(boxes, scores, [0, 4, 9], **KW)
    with pytest.raises(ValueError):
        ops.video_batch(boxes, scores, [0, 4, 4, 10], **KW)


def test_batch_with_a_video_longer_than_the_wave_series_stage():
    """a 2 500-frame video (VID videos run past 2 000 frames; the reference has no length limit, vdet/tubelet_cls.py:284-303,
    :386-414) between two short ones: its tubelet series do not fit the LDS stage of the one-wave-per-series kernel and
    take the one-thread-per-series kernel -- same results as the single-video entry points (whose own long-video path is
    the same serial kernel) and, for the short videos, as ever"""
    frames = [9, 2500, 14]
    B, C = 40, 3
    vids, off, boxes, scores = _videos(frames, B, C, seed=7100)
    # gaps in the long video's tubelet scores: a stretch of frames whose proposals do not overlap the tracked boxes enough
    out = ops.video_batch(boxes, scores, off, overlap_thres=0.6, window=3, **KW)
    one = _lib.Context(torch.cuda.current_device())
    one.set_cache(True)
    for v, (b, s) in enumerate(vids):
        tb, ts = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
        ki, kc, tr, an, nt = ops.nms_track_volume(tb, ts, ctx=one, **KW)
        det, pool, bx = ops.rescore_tracks(tr, nt, tb, ts, overlap_thres=0.6, window=3, ctx=one)
        assert torch.equal(out["ntracks"][v], nt), v
        for c in range(C):
            n = int(nt[c])
            assert _eq(out["tracks"][v][c, :n], tr[c, :n]), (v, c)
            assert _eq(out["det"][v][c, :n], det[c, :n]) and _eq(out["pooled"][v][c, :n], pool[c, :n]), (v, c)
            assert _eq(out["tboxes"][v][c, :n], bx[c, :n]), (v, c)
    # the long video's first class against the reference's own series functions restated by the oracle (score completion +
    # temporal max-pool of every tubelet): do_score_completion / score_proto_temporal_maxpool
    v = 1
    det1 = out["det"][v][0].cpu().numpy()
    pool1 = out["pooled"][v][0].cpu().numpy()
    for t in range(int(out["ntracks"][v, 0])):
        has = ~np.isnan(det1[t])
        x = np.concatenate([[-1e5], det1[t][has], [-1e5]])                   # vdet/tubelet_cls.py:399-412, window 3
        assert has.sum() > 0 and np.array_equal(pool1[t][has], np.maximum(np.maximum(x[:-2], x[1:-1]), x[2:])), t
    one.close()
