"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/vdet_oracle.c`` plus numpy restatements of the
array-form numeric cores of the reference's tubelet post-processing
(``vdet/tubelet_cls.py``, ``vdet/video_det.py``, ``vdet/track.py``).  Every
function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker.  Nothing under
``vdetlib_amd/`` imports it.

Parity status: PINNED against golden vectors recorded from the reference itself
(``tests/golden/make_golden.py``; checked by ``tests/test_oracle_golden.py``),
except ``temporal_conv`` whose arithmetic lives in an external Caffe net that is
not part of the reference tree (parity unpinned).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EDIVZERO = -4


def build(force=False):
    """Compile oracle/vdet_oracle.c with gcc (building the checker is not using it)."""
    so = os.path.join(_HERE, "libvdet_oracle.so")
    src = os.path.join(_HERE, "vdet_oracle.c")
    if force or not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvdet_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvdet_oracle.so")
        if not os.path.isfile(so):
            build()
        L = ctypes.CDLL(so)
        i64, f64, vp, ci = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_int
        L.oracle_argsort_desc.argtypes = [vp, i64, i64, vp]
        L.oracle_nms.argtypes = [vp, i64, i64, ci, f64, vp, vp, vp]
        L.oracle_track_det_nms.argtypes = [vp, i64, i64, vp, i64, i64, f64, vp, vp]
        L.oracle_iou_f64.argtypes = [vp, i64, vp, i64, vp]
        L.oracle_iou_f64.restype = None
        L.oracle_nms_volume.argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, i64, f64, ci,
                                        ctypes.c_float, vp, vp, i64]
        L.oracle_nms_volume_mt.argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, i64, f64, ci,
                                           ctypes.c_float, vp, vp, i64, ci]
        L.oracle_temporal_maxpool_f32.argtypes = [vp, vp, i64, i64, ci, ctypes.c_float]
        L.oracle_temporal_conv_f32.argtypes = [vp, vp, i64, i64, vp, ci, ctypes.c_float,
                                               ctypes.c_float]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check(rc):
    if rc == EDIVZERO:
        raise ZeroDivisionError("float division")
    if rc != 0:
        raise ValueError("oracle error %d" % rc)


def argsort_desc(scores):
    """scores.argsort(kind='stable')[::-1] (tie rule of the build)."""
    s = np.ascontiguousarray(scores, dtype=np.float32)
    out = np.empty(s.shape[0], dtype=np.int64)
    _check(lib().oracle_argsort_desc(_p(s), s.shape[0], 1, _p(out)))
    return out


def _nms(dets, thresh, ncols, order):
    d = np.ascontiguousarray(dets, dtype=np.float32)
    if d.ndim != 2 or (d.shape[0] and d.shape[1] < ncols):
        raise ValueError("dets must be [N,%d] float32" % ncols)
    n = d.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    nk = ctypes.c_int64(0)
    o = None
    if order is not None:
        o = np.ascontiguousarray(order, dtype=np.int64)
    ld = d.shape[1] if n else ncols
    _check(lib().oracle_nms(_p(d), n, ld, ncols, float(thresh), _p(o) if o is not None else None,
                            _p(keep), ctypes.byref(nk)))
    return keep[:nk.value].tolist()


def nms(dets, thresh, order=None):
    """utils/nms.pyx:17-68."""
    return _nms(dets, thresh, 5, order)


def vid_nms(dets, thresh, order=None):
    """utils/nms.pyx:71-125."""
    return _nms(dets, thresh, 6, order)


def track_det_nms(tracks, dets, thresh):
    """utils/nms.pyx:128-189."""
    t = np.ascontiguousarray(tracks, dtype=np.float32)
    d = np.ascontiguousarray(dets, dtype=np.float32)
    keep = np.empty(max(d.shape[0], 1), dtype=np.int64)
    nk = ctypes.c_int64(0)
    _check(lib().oracle_track_det_nms(_p(t), t.shape[0], t.shape[1] if t.shape[0] else 5,
                                      _p(d), d.shape[0], d.shape[1] if d.shape[0] else 6,
                                      float(thresh), _p(keep), ctypes.byref(nk)))
    return keep[:nk.value].tolist()


def iou(boxes1, boxes2):
    """utils/common.py:451-468 (float64, +1 convention)."""
    b1 = np.ascontiguousarray(np.asarray(boxes1).astype('float').reshape(-1, 4))
    b2 = np.ascontiguousarray(np.asarray(boxes2).astype('float').reshape(-1, 4))
    out = np.empty((b1.shape[0], b2.shape[0]), dtype=np.float64)
    with np.errstate(all='ignore'):
        lib().oracle_iou_f64(_p(b1), b1.shape[0], _p(b2), b2.shape[0], _p(out))
    return out


def nms_volume(boxes, scores, thresh, score_thresh=None, cap=None, frames=None, classes=None, threads=1):
    """Per-(frame,class) nms over boxes [F,B,4] / scores [F,B,C]: image_det.py:117-123 applied to
    every (frame, class) of video_det.py:89-99's loop.  Returns keep_idx [F,C,cap] (-1 padded),
    keep_cnt [F,C]; only the requested frame/class sub-ranges are filled."""
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    F, B, C = s.shape
    cap = B if cap is None else cap
    f0, f1 = frames if frames is not None else (0, F)
    c0, c1 = classes if classes is not None else (0, C)
    idx = np.full((F, C, cap), -1, dtype=np.int32)
    cnt = np.zeros((F, C), dtype=np.int32)
    if threads > 1:     # independent problems on host threads (bench.py's all-core CPU baseline)
        _check(lib().oracle_nms_volume_mt(_p(b), _p(s), F, B, C, f0, f1, c0, c1, float(thresh),
                                          0 if score_thresh is None else 1,
                                          0.0 if score_thresh is None else float(score_thresh),
                                          _p(idx), _p(cnt), cap, int(threads)))
        return idx, cnt
    _check(lib().oracle_nms_volume(_p(b), _p(s), F, B, C, f0, f1, c0, c1, float(thresh),
                                   0 if score_thresh is None else 1,
                                   0.0 if score_thresh is None else float(score_thresh),
                                   _p(idx), _p(cnt), cap))
    return idx, cnt


def temporal_maxpool(vol, window, pad=-1e5):
    """Array form of score_proto_temporal_maxpool (vdet/tubelet_cls.py:386-414) along axis 0."""
    if window % 2 != 1:
        raise ValueError('Window size must be odd!')
    v = np.ascontiguousarray(vol, dtype=np.float32)
    F = v.shape[0]
    S = int(np.prod(v.shape[1:])) if v.ndim > 1 else 1
    out = np.empty_like(v)
    _check(lib().oracle_temporal_maxpool_f32(_p(v), _p(out), F, S, int(window), float(pad)))
    return out


def temporal_conv(vol, taps, bias=0.0, pad=0.0):
    """Single-channel temporal convolution along axis 0 (build-defined op; parity unpinned:
    the reference's TCN is an external Caffe net, vdet/tubelet_cls.py:15-51)."""
    v = np.ascontiguousarray(vol, dtype=np.float32)
    t = np.ascontiguousarray(taps, dtype=np.float32)
    F = v.shape[0]
    S = int(np.prod(v.shape[1:])) if v.ndim > 1 else 1
    out = np.empty_like(v)
    _check(lib().oracle_temporal_conv_f32(_p(v), _p(out), F, S, _p(t), t.shape[0], float(bias),
                                          float(pad)))
    return out


# ---------------------------------------------------------------------------------------------
# numpy restatements (array forms) of the python-level numeric cores
# ---------------------------------------------------------------------------------------------

def threshold_topk(scores, boxes, thresh=0.05, max_per_image=100):
    """vdet/video_det.py:89-99 for ONE frame: scores [B,C+1], boxes [B,4(C+1)].
    Returns list over classes j (entry 0 = [] for background) of float32 [n,5] arrays."""
    num_classes = scores.shape[1]
    out = [[] for _ in range(num_classes)]
    for j in range(1, num_classes):
        inds = np.where(scores[:, j] > thresh)[0]
        cls_scores = scores[inds, j]
        cls_boxes = boxes[inds, j * 4:(j + 1) * 4]
        if len(cls_scores) > max_per_image:
            # reference uses the default (unstable) argsort; stable here == same on tie-free data
            top_inds = np.argsort(-cls_scores, kind='stable')[:max_per_image]
            cls_scores = cls_scores[top_inds]
            cls_boxes = cls_boxes[top_inds, :]
        out[j] = np.hstack((cls_boxes, cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
    return out


def spatial_maxpool(tubelet_boxes, det_boxes, det_scores, overlap_thres=0.7):
    """Array form of the inner loop of raw_dets_spatial_max_pooling / dets_spatial_max_pooling
    (vdet/tubelet_cls.py:514-532, :327-347) for ONE frame: for each tubelet box pick, among the
    dets with iou > overlap_thres (strict, float64), the first arg-max of the class score.
    Returns (det_score float64 [T], bbox float64 [T,4], hit bool [T]); misses get -1e5 and keep
    their box."""
    tb = np.asarray(tubelet_boxes, dtype=np.float64).reshape(-1, 4)
    db = np.asarray(det_boxes)
    sc = np.asarray(det_scores).ravel()
    T = tb.shape[0]
    out_s = np.full(T, -1e5, dtype=np.float64)
    out_b = tb.copy()
    hit = np.zeros(T, dtype=bool)
    for t in range(T):
        overlaps = iou([tb[t]], db)
        overlap_idx = (overlaps > overlap_thres).ravel()
        if np.any(overlap_idx):
            conf_boxes = db[overlap_idx]
            conf_scores = sc[overlap_idx]
            max_idx = np.argmax(conf_scores)
            out_s[t] = float(conf_scores[max_idx])
            out_b[t] = np.asarray(conf_boxes[max_idx], dtype=np.float64)
            hit[t] = True
    return out_s, out_b, hit


def score_completion(det_scores):
    """Array form of do_score_completion (vdet/tubelet_cls.py:284-303) for ONE tubelet:
    runs of det_score <= -10 are filled by edge extension / linear interpolation.  Returns a new
    float64 array.  Raises IndexError where the reference does (whole tubelet missing)."""
    s = [float(x) for x in det_scores]
    n = len(s)
    for i in range(n):
        if s[i] > -10:
            continue
        j = i
        while j < n and s[j] <= -10:
            j += 1
        if i == 0:
            if j == n:
                raise IndexError('list index out of range')
            for k in range(i, j):
                s[k] = s[j]
        elif j == n:
            for k in range(i, j):
                s[k] = s[i - 1]
        else:
            l, r = s[i - 1], s[j]
            for k in range(i, j):
                s[k] = l + (r - l) * (k - i + 1) / (j - i + 1)
    return np.asarray(s, dtype=np.float64)


def interp_linear(x, y, x_new):
    """scipy.interpolate.interp1d(kind='linear') as evaluated by the scipy of this image
    (1.15: 1-D int/float64 data is delegated to numpy.interp): exact y at a knot, otherwise
    slope = (y[j+1]-y[j])/(x[j+1]-x[j]); y = slope*(x_new-x[j]) + y[j] with x[j] the left knot;
    plus the one-step linear extrapolation of extrap1d (vdet/tubelet_cls.py:416-428).
    (The py2-era scipy evaluated knots through the previous interval; the two differ by <= 1 ulp,
    inside the 1e-5 float tolerance of the north star.)"""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    out = np.empty(len(x_new), dtype=np.float64)
    for n, xv in enumerate(x_new):
        xv = float(xv)
        if xv < x[0]:
            out[n] = y[0] + (xv - x[0]) * (y[1] - y[0]) / (x[1] - x[0])
        elif xv > x[-1]:
            out[n] = y[-1] + (xv - x[-1]) * (y[-1] - y[-2]) / (x[-1] - x[-2])
        else:
            j = int(np.searchsorted(x, xv, side='right')) - 1
            if j == len(x) - 1 or x[j] == xv:
                out[n] = y[j]
            else:
                slope = (y[j + 1] - y[j]) / (x[j + 1] - x[j])
                out[n] = slope * (xv - x[j]) + y[j]
    return out


def tubelet_interpolation(frames, fields, max_frames):
    """Array form of score_proto_interpolation for ONE tubelet (vdet/tubelet_cls.py:453-487):
    frames [L] ints; fields [L,K] (x1,y1,x2,y2,det_score,anchor).  Returns (dense_frames, dense
    fields [L',K] float64)."""
    frames = [int(f) for f in frames]
    fields = np.asarray(fields, dtype=np.float64)
    min_idx, max_idx = min(frames), max(frames)
    if min_idx == 2:
        min_idx = 1
    if max_idx == max_frames - 1:
        max_idx = max_frames
    dense = list(range(min_idx, max_idx + 1))
    cols = [interp_linear(frames, fields[:, k], dense) for k in range(fields.shape[1])]
    return np.asarray(dense, dtype=np.int64), np.stack(cols, axis=1)


# ---------------------------------------------------------------------------------------------
# greedy tubelet generation on arrays (vdet/track.py:189-252) with the build's IoU-linking tracker
# ---------------------------------------------------------------------------------------------

def _iou_f32_row(cur, boxes):
    """float32 IoU of one box (as the "i" box) with boxes [B,4], utils/nms.pyx:57-64 arithmetic."""
    f = np.float32
    cur = cur.astype(f)
    b = boxes.astype(f)
    with np.errstate(all='ignore'):
        xx1 = np.where(cur[0] >= b[:, 0], cur[0], b[:, 0])
        yy1 = np.where(cur[1] >= b[:, 1], cur[1], b[:, 1])
        xx2 = np.where(cur[2] <= b[:, 2], cur[2], b[:, 2])
        yy2 = np.where(cur[3] <= b[:, 3], cur[3], b[:, 3])
        w = (xx2 - xx1) + f(1)
        w = np.where(f(0) >= w, f(0), w)
        h = (yy2 - yy1) + f(1)
        h = np.where(f(0) >= h, f(0), h)
        inter = w * h
        carea = ((cur[2] - cur[0]) + f(1)) * ((cur[3] - cur[1]) + f(1))
        areas = ((b[:, 2] - b[:, 0]) + f(1)) * ((b[:, 3] - b[:, 1]) + f(1))
        return (inter / ((carea + areas) - inter)).astype(f)


def iou_link_rows(boxes, anchor_frame0, anchor_box, link_thres=0.5, max_frames=0):
    """The build's tracker plug-in on arrays (parity UNPINNED: the reference's trackers are external
    MATLAB code).  boxes [F,B,4]; returns rows [F,5] (x1,y1,x2,y2,score), NaN where no box."""
    return iou_link_rows_box(boxes, anchor_frame0, np.trunc(boxes[anchor_frame0, anchor_box]), link_thres, max_frames)


def iou_link_rows_box(boxes, anchor_frame0, anchor_bbox, link_thres=0.5, max_frames=0):
    """iou_link_rows from the (int-truncated) anchor BOX, the form a ``track_method`` plug-in receives
    (vdet/track.py:225-226: track_method(vid_proto, anchor_frame_id, map(int, bbox), opts)).
    tests/golden/make_golden.py hands exactly this function (+ the reference's own
    tracks_proto_from_boxes) to the reference's greedily_track_from_raw_dets to pin the LINK stage."""
    F = boxes.shape[0]
    rows = np.full((F, 5), np.nan, dtype=np.float32)
    anchor = np.asarray(anchor_bbox, dtype=np.float32)
    rows[anchor_frame0] = [anchor[0], anchor[1], anchor[2], anchor[3], 1.0]
    reach = F if max_frames <= 0 else int(np.ceil((max_frames + 1) / 2.)) - 1
    for direction in (1, -1):
        cur = anchor
        for step in range(1, reach + 1):
            f = anchor_frame0 + direction * step
            if f < 0 or f >= F:
                break
            ious = _iou_f32_row(cur, boxes[f])
            ok = ~np.isnan(ious)
            if not ok.any():
                break
            j = int(np.argmax(np.where(ok, ious, np.float32(-1))))
            if not (float(ious[j]) >= link_thres):
                break
            cur = np.trunc(boxes[f, j]).astype(np.float32)
            rows[f] = [cur[0], cur[1], cur[2], cur[3], ious[j]]
    return rows


def greedy_track_volume(boxes, scores_c, nms_thres=0.3, thres=0.0, max_tracks=10, link_thres=0.5, max_frames=0):
    """vdet/track.py:189-252 for ONE class on arrays: boxes [F,B,4] f32, scores_c [F,B] f32.
    Returns (tracks [max_tracks,F,5], anchors [max_tracks,3], ntracks)."""
    F, B = scores_c.shape
    frame = np.repeat(np.arange(1, F + 1), B).astype(np.float64)
    det = np.hstack([frame[:, None], boxes.reshape(-1, 4).astype(np.float64),
                     scores_c.reshape(-1, 1).astype(np.float64)])
    order = np.argsort(-det[:, 5], kind='stable')           # :200 sorted(..., reverse=True) is stable
    det_info = det[order].astype(np.float32)
    ids_by_frame = {}
    for i, fr in enumerate(det_info[:, 0]):
        ids_by_frame.setdefault(int(fr), []).append(i)
    keep = np.ones(len(det_info), dtype=bool)
    cur = 0
    tracks = np.full((max_tracks, F, 5), np.nan, dtype=np.float32)
    anchors = np.zeros((max_tracks, 3), dtype=np.float32)
    nt = 0
    while keep.any() and nt < max_tracks:
        while cur < len(keep) and not keep[cur]:
            cur += 1
        if cur == len(keep):
            break
        top = det_info[cur]
        top_flat = order[cur]
        cur += 1
        if top[-1] < thres:
            break
        af0 = int(top[0]) - 1
        rows = iou_link_rows(boxes, af0, int(top_flat - af0 * B), link_thres, max_frames)
        tracks[nt] = rows
        anchors[nt] = [af0 + 1, top_flat - af0 * B, top[-1]]
        nt += 1
        for f in range(F):
            if np.isnan(rows[f, 0]):
                continue
            det_ids = [i for i in ids_by_frame.get(f + 1, []) if keep[i]]
            if not det_ids:
                continue
            t = np.asarray([[f + 1, rows[f, 0], rows[f, 1], rows[f, 2], rows[f, 3]]], dtype=np.float32)
            kp = set(track_det_nms(t, det_info[det_ids], nms_thres))
            for i, det_id in enumerate(det_ids):
                if i not in kp:
                    keep[det_id] = False
    return tracks, anchors, nt


def tcn_forward(x, layers):
    """The build's tubelet TCN (vdetlib_amd/vdet/tcn.py) on the CPU, same accumulation order:
    x [Cin,L] f32; layers = [(W [Cout,Cin,K], b [Cout])...]; ReLU between layers, channel softmax at
    the end.  Parity unpinned (the reference's net is external)."""
    f = np.float32
    x = np.asarray(x, dtype=f)
    for li, (w, b) in enumerate(layers):
        w = np.asarray(w, dtype=f); b = np.asarray(b, dtype=f)
        Cout, Cin, K = w.shape
        L = x.shape[1]
        h = K // 2
        xp = np.zeros((Cin, L + 2 * h), dtype=f)
        xp[:, h:h + L] = x
        out = np.repeat(b[:, None], L, 1).astype(f)
        for ci in range(Cin):
            for k in range(K):
                out = (out + (w[:, ci, k][:, None] * xp[ci, k:k + L][None, :]).astype(f)).astype(f)
        if li < len(layers) - 1:
            out = np.where(out > 0, out, f(0)).astype(f)
        x = out
    m = x.max(0)
    e = np.exp((x - m).astype(f)).astype(f)
    ssum = np.zeros(x.shape[1], dtype=f)
    for c in range(x.shape[0]):
        ssum = (ssum + e[c]).astype(f)
    return (e / ssum).astype(f)


def rescored_tubelets(boxes, scores, nms_thres, thres, max_tracks, link_thres, pool_thres, window, max_frames=0,
                      return_det=False):
    """greedy_track_volume for every class, then raw_dets_spatial_max_pooling + do_score_completion +
    score_proto_temporal_maxpool(window) of every tubelet (vdet/track.py:189-252,
    vdet/tubelet_cls.py:493-535, :284-303, :386-414) -- the array form of what
    vdet_track_volume + vdet_rescore_tracks return: tracks [C,T,F,5], ntracks [C],
    pooled score [C,T,F] f64, regressed box [C,T,F,4] (NaN where a track has no box)."""
    F, B, C = scores.shape
    T = max_tracks
    h = window // 2
    wtr = np.full((C, T, F, 5), np.nan, np.float32)
    wnt = np.zeros(C, np.int32)
    wsc = np.full((C, T, F), np.nan)
    wdet = np.full((C, T, F), np.nan)
    wbx = np.full((C, T, F, 4), np.nan, np.float32)
    for c in range(C):
        t_, a_, n_ = greedy_track_volume(boxes, scores[:, :, c], nms_thres, thres, T, link_thres, max_frames)
        wtr[c], wnt[c] = t_, n_
        for t in range(n_):
            fr = [f for f in range(F) if not np.isnan(t_[t, f, 0])]
            s, bx = [], []
            for f in fr:
                ss, bb, _ = spatial_maxpool([t_[t, f, :4]], boxes[f], scores[f, :, c], pool_thres)
                s.append(ss[0])
                bx.append(bb[0])
            comp = score_completion(s)
            pool = [max(comp[g] if 0 <= g < len(comp) else -1e5 for g in range(i - h, i + h + 1)) for i in range(len(comp))]
            wsc[c, t, fr] = pool
            wdet[c, t, fr] = comp
            wbx[c, t, fr] = np.asarray(bx, np.float32)
    if return_det:
        return wtr, wnt, wsc, wbx, wdet
    return wtr, wnt, wsc, wbx
