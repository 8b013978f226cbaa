"""-m gpu: the asynchronous video step (vdet_set_async: no host synchronisation inside the volume entry
points) gives the results of the synchronous one; the advisor's two cache-soundness cases."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _ctx():
    import torch
    from vdetlib_amd import _lib
    return _lib.Context(torch.cuda.current_device())


def _same(a, b):
    import torch
    return all(torch.equal(x.nan_to_num(-7.0), y.nan_to_num(-7.0)) for x, y in zip(a, b))


def test_async_equals_sync_over_several_videos():
    import torch
    from vdetlib_amd import ops
    cx = _ctx()
    cx.set_cache(True)
    vids = [synth.coherent_video(6000 + i, 12, 600, 6) for i in range(4)]
    vids[2][0][3, 10] = [5, 5, 5, np.nan]          # one irregular frame: the eager suppress kernel must run
    want = []
    for b, s in vids:
        tb, ts = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
        want.append(ops.nms_track_volume(tb, ts, thres=0.3, max_tracks=5) +
                    ops.rescore_tracks(*ops.track_volume(tb, ts, thres=0.3, max_tracks=5)[::2], tb, ts))
    cx.set_async(True)
    got = []
    keep = []
    for rep in range(2):            # the first video still builds synchronously (pool size unknown), the rest do not
        for b, s in vids:
            tb, ts = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
            keep.append((tb, ts))
            cx.invalidate()
            r = ops.nms_track_volume(tb, ts, thres=0.3, max_tracks=5, sync=False, ctx=cx)
            r2 = ops.rescore_tracks(r[2], r[4], tb, ts, sync=False, ctx=cx)
            got.append(r + r2)
    cx.sync()
    for i, g in enumerate(got):
        w = want[i % len(vids)]
        assert _same(w[:5], g[:5]), i
        assert _same(w[5:], g[5:]), i
    cx.close()


def test_async_pool_overflow_is_reported_not_silent():
    """A graph that outgrows the pool sized from earlier videos: vdet_sync says so (RetryError), and the retry works."""
    import torch
    from vdetlib_amd import ops, _lib
    cx = _ctx()
    cx.set_cache(True)
    cx.set_async(True)
    F, B, C = 4, 3000, 2
    rng = np.random.RandomState(1)
    sparse = np.stack([synth.boxes_1(rng, B) for _ in range(F)], 0)
    dense = sparse.copy()
    dense[:, :, :2] = dense[:, :, :2] % 60          # everything piled into one corner: ~B^2/2 edges per frame
    dense[:, :, 2:] = dense[:, :, :2] + 200
    sc = rng.rand(F, B, C).astype(np.float32)
    ts = torch.from_numpy(sc).cuda()
    t_sparse, t_dense = torch.from_numpy(sparse).cuda(), torch.from_numpy(dense).cuda()
    ops.nms_volume(t_sparse, ts, 0.3, ctx=cx)       # synchronous first build: learns the pool size of a sparse graph
    cx.invalidate()
    idx, cnt = ops.nms_volume(t_dense, ts, 0.3, sync=False, ctx=cx)
    with pytest.raises(_lib.RetryError):
        cx.sync()
    cx.invalidate()
    idx, cnt = ops.nms_volume(t_dense, ts, 0.3, sync=False, ctx=cx)
    cx.sync()
    widx, wcnt = ops.nms_volume(t_dense, ts, 0.3)   # default context, synchronous
    assert torch.equal(cnt, wcnt) and torch.equal(idx, widx)
    cx.close()
    # ... and an op called with sync=True repeats the enqueue by itself
    c2 = _ctx()
    c2.set_cache(True)
    c2.set_async(True)
    ops.nms_volume(t_sparse, ts, 0.3, ctx=c2)
    c2.invalidate()
    idx2, cnt2 = ops.nms_volume(t_dense, ts, 0.3, ctx=c2)
    assert torch.equal(cnt2, wcnt) and torch.equal(idx2, widx)
    c2.close()


def test_track_volume_beyond_the_index_limit_after_a_smaller_video(oracle):
    """Advisor (round 1): 17408 < B <= 18432 takes neither K0 nor the frame index; a context that served a
    B = 10000 video before must not hand the tracking kernels that video's stale index / flags."""
    import torch
    from vdetlib_amd import ops
    cx = _ctx()
    cx.set_cache(True)
    b1, s1 = synth.coherent_video(6100, 3, 10000, 1)
    ops.track_volume(torch.from_numpy(b1).cuda(), torch.from_numpy(s1).cuda(), thres=0.5, max_tracks=2, ctx=cx)
    F, B = 3, 17500
    b2, s2 = synth.coherent_video(6101, F, B, 1)
    tr, an, nt = ops.track_volume(torch.from_numpy(b2).cuda(), torch.from_numpy(s2).cuda(), thres=0.5, max_tracks=2, ctx=cx)
    wt, wa, wn = oracle.greedy_track_volume(b2, s2[:, :, 0], 0.3, 0.5, 2, 0.5, 0)
    assert int(nt[0]) == wn
    assert np.array_equal(tr[0, :wn].cpu().numpy(), wt[:wn], equal_nan=True)
    tb2, ts2 = torch.from_numpy(b2).cuda(), torch.from_numpy(s2).cuda()
    det, pooled, ob = ops.rescore_tracks(tr, nt, tb2, ts2, ctx=cx)
    wtr, wnt, wsc, wbx = oracle.rescored_tubelets(b2, s2, 0.3, 0.5, 2, 0.5, 0.7, 3)
    np.testing.assert_allclose(pooled[0, :wn].cpu().numpy(), wsc[0, :wn], rtol=0, atol=1e-9, equal_nan=True)
    cx.close()


def test_host_entry_points_do_not_poison_the_cache(oracle):
    """Advisor (round 1): with the cache on, two host-buffer calls of the same size must not share an x-index,
    and a host call between two volume calls must not leave the volume path a clobbered graph."""
    import torch
    from vdetlib_amd import ops, _lib
    from vdetlib_amd.utils import cython_nms
    cx = _lib.get_context(torch.cuda.current_device())
    cx.set_cache(True)
    try:
        d1 = synth.dets5(6200, 700)
        d2 = synth.dets5(6201, 700)
        assert cython_nms.nms(d1, 0.3) == oracle.nms(d1, 0.3)
        assert cython_nms.nms(d2, 0.3) == oracle.nms(d2, 0.3)
        boxes, scores = synth.video(6202, 3, 700, 2)
        tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        idx, cnt = ops.nms_volume(tb, ts, 0.3)
        assert cython_nms.nms(d1, 0.3) == oracle.nms(d1, 0.3)        # overwrites the shared scratch
        idx2, cnt2 = ops.nms_volume(tb, ts, 0.3)                      # same pointers: must rebuild, not reuse
        widx, wcnt = oracle.nms_volume(boxes, scores, 0.3)
        assert np.array_equal(cnt2.cpu().numpy(), wcnt) and np.array_equal(idx2.cpu().numpy(), widx)
        assert torch.equal(idx, idx2)
    finally:
        cx.set_cache(False)


def test_async_step_never_waits_for_the_device():
    """vdet_query(ctx, 8) counts every hipStreamSynchronize the library issues: once a context has built one graph,
    an asynchronous step (volume pass + NMS + tubelets + re-scoring of a NEW video) adds none."""
    import torch
    from vdetlib_amd import ops
    cx = _ctx()
    cx.set_cache(True)
    cx.set_async(True)
    vids = [synth.coherent_video(6300 + i, 10, 500, 8) for i in range(4)]
    dev = [(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()) for b, s in vids]

    def step(tb, ts):
        cx.invalidate()
        pooled, conv = ops.volume_pass(ts, 3, [0.25, 0.5, 0.25], ctx=cx)
        r = ops.nms_track_volume(tb, ts, thres=0.3, max_tracks=4, sync=False, ctx=cx, pad=False)
        rr = ops.rescore_tracks(r[2], r[4], tb, ts, sync=False, ctx=cx)
        return r + rr + (pooled, conv)

    step(*dev[0])                 # the first graph of a context is built synchronously (learns the scratch size)
    cx.sync()
    before = cx.query(8)
    outs = [step(tb, ts) for tb, ts in dev[1:]]
    assert cx.query(8) == before, "an asynchronous step waited for the device"
    cx.sync()
    assert cx.query(8) == before + 1
    for (tb, ts), got in zip(dev[1:], outs):      # and the results are the synchronous ones
        want = ops.nms_track_volume(tb, ts, thres=0.3, max_tracks=4)
        for a, b in zip(want[1:], got[1:5]):
            assert torch.equal(a.nan_to_num(-7.0), b.nan_to_num(-7.0))
    cx.close()
