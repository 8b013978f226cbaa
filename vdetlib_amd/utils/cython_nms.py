"""Drop-in for the reference's only native module ``vdetlib.utils.cython_nms``
(built from utils/nms.pyx by setup.py:8-14; imported by vdet/image_det.py:9, vdet/video_det.py:11,
vdet/track.py:13).  Same three functions, same argument meaning, same return type (a python list
of python ints in descending score order), same errors -- computed by the gfx950 kernels behind
``vdet_nms_f32`` / ``vdet_track_det_nms_f32`` (include/vdet_hip.h).  No CPU fallback.
"""
import ctypes

import numpy as np

from .. import _lib


def _as_f32_2d(a, name, ncols):
    # Cython signature np.ndarray[np.float32_t, ndim=2]: anything else is rejected
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % a.ndim)
    if a.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t' but got '%s'" % a.dtype.name)
    if a.shape[1] < ncols:
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (ncols - 1, a.shape[1]))
    # the buffer interface honours arbitrary strides; the C-ABI wants a row stride in elements
    if a.strides[1] != 4 or a.strides[0] % 4 != 0 or a.strides[0] < 0:
        a = np.ascontiguousarray(a)
    return a


def _thresh(thresh):
    if thresh is None:
        raise TypeError("a float is required")
    return float(thresh)


def _run_nms(dets, thresh, ncols, order):
    d = _as_f32_2d(dets, 'dets', ncols)
    t = _thresh(thresh)
    n = d.shape[0]
    if n == 0:
        return []
    ctx = _lib.get_context()
    ctx.reset_stream()          # host-buffer entry points are synchronous on the context's own stream
    keep = np.empty(n, dtype=np.int64)
    nk = ctypes.c_int64(0)
    o = None
    if order is not None:
        o = np.ascontiguousarray(order, dtype=np.int64)
        if o.shape != (n,):
            raise ValueError("order must have one entry per detection")
    ld = d.strides[0] // 4 if n > 1 else d.shape[1]
    ctx.check(ctx.lib.vdet_nms_f32(ctx.h, d.ctypes.data, n, ld, ncols, t,
                                   o.ctypes.data if o is not None else None,
                                   keep.ctypes.data, ctypes.byref(nk)))
    return keep[:nk.value].tolist()


def nms(dets, thresh, order=None):
    """utils/nms.pyx:17-68 -- dets float32 [N,5] (x1,y1,x2,y2,score).

    ``order`` (extension): the permutation the reference's ``scores.argsort()[::-1]`` produced, to
    reproduce a specific machine's tie order; default = descending score, ties by descending index."""
    return _run_nms(dets, thresh, 5, order)


def vid_nms(dets, thresh, order=None):
    """utils/nms.pyx:71-125 -- dets float32 [N,6] (frame,x1,y1,x2,y2,score)."""
    return _run_nms(dets, thresh, 6, order)


def track_det_nms(tracks, dets, thresh):
    """utils/nms.pyx:128-189 -- tracks float32 [T,5] (frame,x1,y1,x2,y2), dets float32 [M,6]."""
    t = _as_f32_2d(tracks, 'tracks', 5)
    d = _as_f32_2d(dets, 'dets', 6)
    th = _thresh(thresh)
    m = d.shape[0]
    if m == 0:
        return []
    ctx = _lib.get_context()
    ctx.reset_stream()
    keep = np.empty(m, dtype=np.int64)
    nk = ctypes.c_int64(0)
    ldt = t.strides[0] // 4 if t.shape[0] > 1 else max(t.shape[1], 5)
    ldd = d.strides[0] // 4 if m > 1 else d.shape[1]
    ctx.check(ctx.lib.vdet_track_det_nms_f32(ctx.h, t.ctypes.data if t.shape[0] else None, t.shape[0], ldt,
                                             d.ctypes.data, m, ldd, th, keep.ctypes.data, ctypes.byref(nk)))
    return keep[:nk.value].tolist()


def track_det_nms_batch(tracks, dets, offsets, thresh, track_offsets=None):
    """K independent ``track_det_nms`` problems in ONE call (extension; include/vdet_hip.h: vdet_track_det_nms_batch) -- the
    reference's per-tracked-box pattern of vdet/track.py:236-250, where the boxes of a tracklet sit on different frames.
    Problem k = ``track_det_nms(tracks[to[k]:to[k+1]], dets[offsets[k]:offsets[k+1]], thresh)``; without ``track_offsets``
    problem k has the ONE track row ``tracks[k]``.  Returns (keep, counts): int64 arrays; problem k's kept positions (inside
    its own rows, descending score) are ``keep[offsets[k] : offsets[k] + counts[k]]``."""
    t = _as_f32_2d(tracks, 'tracks', 5)
    d = _as_f32_2d(dets, 'dets', 6)
    th = _thresh(thresh)
    off = np.ascontiguousarray(offsets, dtype=np.int64).reshape(-1)
    K = off.size - 1
    if K < 0 or off[0] != 0 or off[-1] != d.shape[0] or np.any(np.diff(off) < 0):
        raise ValueError("offsets must run 0 = o[0] <= o[1] <= ... <= o[K] = len(dets)")
    toff = None
    if track_offsets is not None:
        toff = np.ascontiguousarray(track_offsets, dtype=np.int64).reshape(-1)
        if toff.size != K + 1 or toff[0] != 0 or toff[-1] != t.shape[0] or np.any(np.diff(toff) < 0):
            raise ValueError("track_offsets must run 0 = o[0] <= ... <= o[K] = len(tracks)")
    elif t.shape[0] != K:
        raise ValueError("one track row per problem (or pass track_offsets)")
    keep = np.empty(max(d.shape[0], 1), dtype=np.int64)
    counts = np.zeros(max(K, 1), dtype=np.int64)
    if K == 0 or d.shape[0] == 0:
        return keep[:0] if d.shape[0] == 0 else keep, counts[:K]
    ctx = _lib.get_context()
    ctx.reset_stream()
    ldt = t.strides[0] // 4 if t.shape[0] > 1 else max(t.shape[1], 5)
    ldd = d.strides[0] // 4 if d.shape[0] > 1 else d.shape[1]
    ctx.check(ctx.lib.vdet_track_det_nms_batch(ctx.h, t.ctypes.data if t.shape[0] else None, toff.ctypes.data if toff is not None else None,
                                               ldt, d.ctypes.data, off.ctypes.data, K, ldd, th, keep.ctypes.data, counts.ctypes.data))
    return keep, counts[:K]
