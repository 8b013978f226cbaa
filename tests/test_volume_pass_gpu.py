"""-m gpu: vdet_volume_pass (one read of the score volume: temporal max-pool + temporal convolution +
class-major sort keys) against the oracle's temporal ops, and the keys it leaves in the context against
the keys the separate transpose produces (same NMS survivors / tubelets either way)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

TAPS = [0.25, 0.5, 0.25]


def _ctx():
    import torch
    from vdetlib_amd import _lib
    return _lib.Context(torch.cuda.current_device())


@pytest.mark.parametrize("shape", [(7, 300, 8), (5, 100, 200), (12, 37, 12), (3, 1030, 4), (9, 64, 64), (4, 33, 200),
                                   (1, 50, 16), (2, 17, 400)])
@pytest.mark.parametrize("window", [3, 5])
def test_volume_pass_temporal_outputs(oracle, shape, window):
    import torch
    from vdetlib_amd import ops
    F, B, C = shape
    rng = np.random.RandomState(F * 1000 + B + C)
    scores = rng.randn(F, B, C).astype(np.float32)
    scores[0, 0, 0] = np.nan
    if F > 2:
        scores[F // 2, B // 2, C - 1] = np.nan
        scores[F - 1, B - 1, 1] = -np.inf
    taps = TAPS if window == 3 else [0.1, 0.2, 0.4, 0.2, 0.1]
    ts = torch.from_numpy(scores).cuda()
    pooled, conv = ops.volume_pass(ts, window, taps, bias=0.125, pad_conv=0.5)
    assert np.array_equal(pooled.cpu().numpy(), oracle.temporal_maxpool(scores, window), equal_nan=True)
    want = oracle.temporal_conv(scores, taps, 0.125, 0.5)
    got = conv.cpu().numpy()
    # same f32 operation order on both sides -> equal bits (NaN / inf positions included)
    assert np.array_equal(got, want, equal_nan=True)
    pooled2, none = ops.volume_pass(ts, window)
    assert none is None and torch.equal(pooled2.nan_to_num(7.0), pooled.nan_to_num(7.0))


@pytest.mark.parametrize("shape", [(6, 300, 8), (3, 1000, 200), (4, 77, 12)])
@pytest.mark.parametrize("score_thresh", [None, 0.4])
def test_volume_pass_keys_serve_the_nms(oracle, shape, score_thresh):
    """nms_volume after volume_pass (cache on: the transpose is skipped, the pass's keys are used) ==
    nms_volume alone == oracle."""
    import torch
    from vdetlib_amd import ops
    F, B, C = shape
    boxes, scores = synth.video(4100 + B, F, B, C)
    if score_thresh is None:
        scores[1, 5, 3] = np.nan          # NaN sorts first (tie rule) -- through both key paths
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    cx = _ctx()
    cx.set_cache(True)
    cx.set_timing(1)
    ops.volume_pass(ts, 3, TAPS, score_thresh=score_thresh, ctx=cx)
    idx, cnt = ops.nms_volume(tb, ts, 0.3, score_thresh=score_thresh, ctx=cx)
    t = cx.last_timing()
    assert t["transpose_keys"][1] == 0, "the key transpose ran although volume_pass left keys"
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3, score_thresh)
    assert np.array_equal(cnt.cpu().numpy(), wcnt) and np.array_equal(idx.cpu().numpy(), widx)
    # a different threshold must NOT reuse those keys
    idx2, cnt2 = ops.nms_volume(tb, ts, 0.3, score_thresh=0.9, ctx=cx)
    widx2, wcnt2 = oracle.nms_volume(boxes, scores, 0.3, 0.9)
    assert np.array_equal(cnt2.cpu().numpy(), wcnt2) and np.array_equal(idx2.cpu().numpy(), widx2)
    cx.close()


def test_volume_pass_then_tracking_identical(oracle):
    import torch
    from vdetlib_amd import ops
    boxes, scores = synth.coherent_video(4201, 10, 400, 8)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    ref = ops.nms_track_volume(tb, ts, thres=0.5, max_tracks=4)
    cx = _ctx()
    cx.set_cache(True)
    pooled, conv = ops.volume_pass(ts, 3, TAPS, ctx=cx)
    got = ops.nms_track_volume(tb, ts, thres=0.5, max_tracks=4, ctx=cx)
    for a, b in zip(ref, got):
        assert torch.equal(a.nan_to_num(-7.0), b.nan_to_num(-7.0))
    assert np.array_equal(pooled.cpu().numpy(), oracle.temporal_maxpool(scores, 3))
    cx.invalidate()             # a rewritten buffer: nothing may be reused
    ts2 = ts.clone()
    ts.copy_(torch.rand_like(ts))
    got2 = ops.nms_track_volume(tb, ts, thres=0.5, max_tracks=4, ctx=cx)
    ref2 = ops.nms_track_volume(tb, ts.clone(), thres=0.5, max_tracks=4)
    for a, b in zip(ref2, got2):
        assert torch.equal(a.nan_to_num(-7.0), b.nan_to_num(-7.0))
    del ts2
    cx.close()


def test_volume_pass_fallback_shapes(oracle):
    """C % 4 != 0 / window 7: the separate kernels run, results unchanged, no keys are left behind."""
    import torch
    from vdetlib_amd import ops
    boxes, scores = synth.video(4300, 5, 120, 30)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    cx = _ctx()
    cx.set_cache(True)
    pooled, conv = ops.volume_pass(ts, 3, TAPS, ctx=cx)
    assert np.array_equal(pooled.cpu().numpy(), oracle.temporal_maxpool(scores, 3))
    assert np.array_equal(conv.cpu().numpy(), oracle.temporal_conv(scores, TAPS, 0.0, 0.0))
    idx, cnt = ops.nms_volume(tb, ts, 0.3, ctx=cx)
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3)
    assert np.array_equal(cnt.cpu().numpy(), wcnt) and np.array_equal(idx.cpu().numpy(), widx)
    s8 = torch.from_numpy(np.ascontiguousarray(scores[:, :, :8])).cuda()
    p7, _ = ops.volume_pass(s8, 7, ctx=cx)
    assert np.array_equal(p7.cpu().numpy(), oracle.temporal_maxpool(scores[:, :, :8], 7))
    cx.close()


@pytest.mark.parametrize("C", [200, 40, 12])
def test_volume_pass_tilings_agree(oracle, C):
    """The tiling of the pass follows the class count (512 threads x 32 boxes at C = 200, 256 x 64 at C = 40, ...): every one
    gives the same three outputs as the oracle."""
    import torch
    from vdetlib_amd import ops
    F, B = 6, 211
    boxes, scores = synth.video(4400, F, B, C)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    cx = _ctx()
    cx.set_cache(True)
    pooled, conv = ops.volume_pass(ts, 3, TAPS, ctx=cx)
    idx, cnt = ops.nms_volume(tb, ts, 0.3, ctx=cx)
    assert np.array_equal(pooled.cpu().numpy(), oracle.temporal_maxpool(scores, 3))
    assert np.array_equal(conv.cpu().numpy(), oracle.temporal_conv(scores, TAPS, 0.0, 0.0))
    widx, wcnt = oracle.nms_volume(boxes, scores, 0.3)
    assert np.array_equal(cnt.cpu().numpy(), wcnt) and np.array_equal(idx.cpu().numpy(), widx)
    cx.close()
