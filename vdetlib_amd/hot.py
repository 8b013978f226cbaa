"""numpy front-ends of the host-buffer C-ABI entry points used by the dict-level API
(vdetlib_amd.vdet.*).  Pure marshalling: every number is produced by a gfx950 kernel."""
import ctypes

import numpy as np

from . import _lib


def _ctx():
    ctx = _lib.get_context()
    ctx.reset_stream()
    return ctx


def _offsets(lengths):
    off = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(np.asarray(lengths, dtype=np.int64), out=off[1:])
    return off


def spatial_maxpool(tub_boxes, tub_group, det_boxes_list, det_scores_list, thres):
    """tub_boxes [T,4], tub_group [T] (index into the per-frame lists), det_boxes_list /
    det_scores_list: one array per frame slot.  Returns (idx int64 [T] (-1 = no overlap),
    score float64 [T])."""
    tb = np.ascontiguousarray(np.asarray(tub_boxes, dtype=np.float64).reshape(-1, 4))
    tg = np.ascontiguousarray(tub_group, dtype=np.int32)
    T = tb.shape[0]
    out_idx = np.full(T, -1, dtype=np.int64)
    out_score = np.full(T, -1e5, dtype=np.float64)
    if T == 0:
        return out_idx, out_score
    off = _offsets([len(b) for b in det_boxes_list])
    M = int(off[-1])
    db = np.zeros((max(M, 1), 4), dtype=np.float64)
    ds = np.zeros(max(M, 1), dtype=np.float64)
    for g, (b, s) in enumerate(zip(det_boxes_list, det_scores_list)):
        if len(b):
            db[off[g]:off[g + 1]] = np.asarray(b).astype('float').reshape(-1, 4)
            ds[off[g]:off[g + 1]] = np.asarray(s, dtype=np.float64).ravel()
    ctx = _ctx()
    ctx.check(ctx.lib.vdet_spatial_maxpool_f64(ctx.h, tb.ctypes.data, tg.ctypes.data, T, db.ctypes.data,
                                               ds.ctypes.data, off.ctypes.data, len(det_boxes_list), float(thres),
                                               out_idx.ctypes.data, out_score.ctypes.data))
    return out_idx, out_score


def series_completion(series_list):
    """do_score_completion on a list of 1-D score sequences; returns new float64 arrays.
    IndexError where the reference raises it."""
    off = _offsets([len(s) for s in series_list])
    vals = np.zeros(max(int(off[-1]), 1), dtype=np.float64)
    for t, s in enumerate(series_list):
        vals[off[t]:off[t + 1]] = np.asarray(s, dtype=np.float64)
    if len(series_list):
        ctx = _ctx()
        ctx.check(ctx.lib.vdet_series_completion_f64(ctx.h, vals.ctypes.data, off.ctypes.data, len(series_list)))
    return [vals[off[t]:off[t + 1]].copy() for t in range(len(series_list))]


def series_maxpool(series_list, window, pad=-1e5):
    if window % 2 != 1:
        raise ValueError('Window size must be odd!')
    off = _offsets([len(s) for s in series_list])
    n = int(off[-1])
    vals = np.zeros(max(n, 1), dtype=np.float64)
    for t, s in enumerate(series_list):
        vals[off[t]:off[t + 1]] = np.asarray(s, dtype=np.float64)
    out = np.zeros(max(n, 1), dtype=np.float64)
    if n:
        ctx = _ctx()
        ctx.check(ctx.lib.vdet_series_maxpool_f64(ctx.h, vals.ctypes.data, out.ctypes.data, off.ctypes.data,
                                                  len(series_list), int(window), float(pad)))
    return [out[off[t]:off[t + 1]].copy() for t in range(len(series_list))]


def series_interp(knot_x_list, knot_y_list, query_list):
    """Per tubelet: knots x [L] (ascending), y [L,K]; queries [Lq].  Returns list of [Lq,K] float64."""
    T = len(knot_x_list)
    if T == 0:
        return []
    K = int(np.asarray(knot_y_list[0]).shape[1])
    koff = _offsets([len(x) for x in knot_x_list])
    qoff = _offsets([len(q) for q in query_list])
    x = np.zeros(max(int(koff[-1]), 1), dtype=np.float64)
    y = np.zeros(max(int(koff[-1]) * K, 1), dtype=np.float64)
    q = np.zeros(max(int(qoff[-1]), 1), dtype=np.float64)
    for t in range(T):
        L = len(knot_x_list[t])
        x[koff[t]:koff[t + 1]] = np.asarray(knot_x_list[t], dtype=np.float64)
        y[koff[t] * K:koff[t + 1] * K] = np.asarray(knot_y_list[t], dtype=np.float64).reshape(L, K).T.ravel()
        q[qoff[t]:qoff[t + 1]] = np.asarray(query_list[t], dtype=np.float64)
    out = np.zeros(max(int(qoff[-1]) * K, 1), dtype=np.float64)
    ctx = _ctx()
    ctx.check(ctx.lib.vdet_series_interp_f64(ctx.h, x.ctypes.data, y.ctypes.data, koff.ctypes.data, q.ctypes.data,
                                             qoff.ctypes.data, T, K, out.ctypes.data))
    res = []
    for t in range(T):
        Lq = int(qoff[t + 1] - qoff[t])
        res.append(out[qoff[t] * K:qoff[t + 1] * K].reshape(K, Lq).T.copy())
    return res


def threshold_topk(scores, thresh, k, col0=1):
    """scores [B, ncols] float32/float64 of ONE frame.  Returns a list over class columns
    col0..ncols-1 of int32 index arrays (vdet/video_det.py:90-97 selection and order)."""
    s = np.asarray(scores)
    if s.dtype not in (np.float32, np.float64):
        s = s.astype(np.float64)
    s = np.ascontiguousarray(s)
    B, ncols = s.shape
    ncls = ncols - col0
    if ncls <= 0:
        return []
    k = int(k)
    idx = np.zeros((ncls, max(k, 1)), dtype=np.int32)
    cnt = np.zeros(ncls, dtype=np.int32)
    ctx = _ctx()
    ctx.check(ctx.lib.vdet_threshold_topk(ctx.h, s.ctypes.data, 1 if s.dtype == np.float64 else 0, B, ncols, col0,
                                          ncls, float(thresh), k, idx.ctypes.data, cnt.ctypes.data))
    return [idx[c, :cnt[c]].copy() for c in range(ncls)]
