"""Device-resident array forms of the hot path (torch tensors are only the memory / stream
plumbing; every computation is a gfx950 kernel behind the C-ABI of include/vdet_hip.h).

A c2-size video (300 frames x 10k boxes x 200 classes) cannot travel as protocol dicts (SURVEY
8a-a14: the JSON would be many GB); these functions are the array transport the dict-level API in
``vdetlib_amd.vdet`` is built on.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _ctx_for(t, ctx=None):
    """The context to run on: the process-wide one of the tensor's device, or an explicit ``ctx``
    (one context per concurrently used HIP stream: a context owns its scratch buffers).  Either way
    the work is enqueued on torch's CURRENT stream."""
    if not t.is_cuda:
        raise ValueError("expected a CUDA/HIP tensor (vdetlib_amd has no CPU path)")
    if ctx is None:
        ctx = _lib.get_context(t.device.index if t.device.index is not None else torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
    return ctx


def _finish(ctx, launch, sync, reset=None):
    """Enqueue ``launch()``; with ``sync`` wait for it -- and when an asynchronous context (``ctx.set_async``) reports
    that a graph build outgrew its scratch (``_lib.RetryError``: the scratch has been enlarged), restore the output
    buffers' initial state (``reset``) and enqueue once more."""
    launch()
    if sync:
        try:
            ctx.sync()
        except _lib.RetryError:
            ctx.invalidate()
            if reset is not None:
                reset()
            launch()
            ctx.sync()


def _keep_buffer(shape, device, pad):
    if pad:
        return torch.full(shape, -1, dtype=torch.int32, device=device)
    return torch.empty(shape, dtype=torch.int32, device=device)


def nms_volume(boxes, scores, thresh=0.3, score_thresh=None, cap=None, layout="FBC", sync=True, ctx=None, topk=0,
               pad=True):
    """Per-(frame, class) greedy NMS of a whole video (vdet/image_det.py:117-123 applied to every
    frame and class of vdet/video_det.py:89-99; == utils/nms.pyx vid_nms per class).

    boxes [F,B,4] f32, scores [F,B,C] (layout 'FBC', class innermost like zs[B,C]) or [F,C,B] ('FCB').
    score_thresh / topk: the candidate selection of fast_rcnn_det_vid (vdet/video_det.py:89-99:
    score > thresh, then the max_per_image best) on the device, before the NMS.
    Returns (keep_idx int32 [F,C,cap], keep_cnt int32 [F,C]); keep_idx[f,c,:cnt] are box indices in
    descending score order, the rest is -1 (pad=False: left uninitialised, saves one fill of the buffer).
    """
    if boxes.dtype != torch.float32 or scores.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    boxes = boxes.contiguous()
    scores = scores.contiguous()
    F, B = boxes.shape[0], boxes.shape[1]
    if layout == "FBC":
        C = scores.shape[2]
        if tuple(scores.shape) != (F, B, C):
            raise ValueError("scores must be [F,B,C]")
        lay = _lib.LAYOUT_FBC
    elif layout == "FCB":
        C = scores.shape[1]
        if tuple(scores.shape) != (F, C, B):
            raise ValueError("scores must be [F,C,B]")
        lay = _lib.LAYOUT_FCB
    else:
        raise ValueError("layout must be 'FBC' or 'FCB'")
    if boxes.shape[2] != 4:
        raise ValueError("boxes must be [F,B,4]")
    cap = B if cap is None else int(cap)
    ctx = _ctx_for(boxes, ctx)
    keep_idx = _keep_buffer((F, C, cap), boxes.device, pad)
    keep_cnt = torch.zeros((F, C), dtype=torch.int32, device=boxes.device)
    _finish(ctx, lambda: ctx.check(ctx.lib.vdet_nms_volume_topk(
        ctx.h, boxes.data_ptr(), scores.data_ptr(), lay, F, B, C, float(thresh), 0 if score_thresh is None else 1,
        0.0 if score_thresh is None else float(score_thresh), int(topk), keep_idx.data_ptr(), keep_cnt.data_ptr(), cap)), sync,
        reset=(lambda: keep_idx.fill_(-1)) if pad else None)
    return keep_idx, keep_cnt


def det_nms_volume(boxes, scores, score_thresh=0.05, topk=100, nms_thresh=0.3, first_class=1, sync=True, ctx=None,
                   want_dets=True):
    """The Fast R-CNN per-class flow of a whole video on the device: ``fast_rcnn_det_vid``'s per-class loop
    (vdet/video_det.py:89-99: ``scores[:, j] > thresh``, the ``max_per_image`` best, rows ``boxes[inds, 4j:4j+4] | score``)
    followed by ``apply_image_nms`` (vdet/image_det.py:117-123) for every frame and class -- every class suppresses its OWN
    regressed boxes (``nms_volume`` is the class-agnostic-box form).

    boxes [F,B,K,4] or [F,B,4K] f32 (the reference's per-frame ``[B, 4K]`` array), scores [F,B,K] f32; classes below
    ``first_class`` (the background column) are skipped.  ``score_thresh=None``: every box is a candidate.  Returns
      dets     [F,K,topk,5] f32 rows in the reference's row order (``all_boxes[j][i]`` = ``dets[i, j, :det_cnt[i, j]]``),
      sel_idx  [F,K,topk] int32 box index of every row,  det_cnt [F,K] int32,
      keep     [F,K,topk] int32 kept row positions, descending score (``apply_image_nms``'s list), keep_cnt [F,K] int32.
    Entries behind the counts are -1 (dets: NaN)."""
    if boxes.dtype != torch.float32 or scores.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    scores = scores.contiguous()
    if scores.dim() != 3:
        raise ValueError("scores must be [F,B,K]")
    F, B, K = scores.shape
    boxes = boxes.contiguous()
    if tuple(boxes.shape) not in ((F, B, K, 4), (F, B, 4 * K)):
        raise ValueError("boxes must be [F,B,K,4] or [F,B,4K]")
    topk = int(topk)
    ctx = _ctx_for(boxes, ctx)
    dev = boxes.device
    dets = torch.full((F, K, topk, 5), float('nan'), dtype=torch.float32, device=dev) if want_dets else None
    sel = torch.full((F, K, topk), -1, dtype=torch.int32, device=dev)
    keep = torch.full((F, K, topk), -1, dtype=torch.int32, device=dev)
    det_cnt = torch.zeros((F, K), dtype=torch.int32, device=dev)
    keep_cnt = torch.zeros((F, K), dtype=torch.int32, device=dev)

    def reset():
        if dets is not None:
            dets.fill_(float('nan'))
        sel.fill_(-1); keep.fill_(-1)
    _finish(ctx, lambda: ctx.check(ctx.lib.vdet_det_nms_volume(
        ctx.h, boxes.data_ptr(), scores.data_ptr(), F, B, K, int(first_class), 0 if score_thresh is None else 1,
        0.0 if score_thresh is None else float(score_thresh), topk, float(nms_thresh),
        dets.data_ptr() if dets is not None else None, sel.data_ptr(), det_cnt.data_ptr(), keep.data_ptr(), keep_cnt.data_ptr())),
        sync, reset=reset)
    return dets, sel, det_cnt, keep, keep_cnt


def nms_volume_ordered(boxes, order, ncand, thresh=0.3, cap=None, ctx=None, pad=True):
    """``nms_volume`` walking the CALLER's lists: order int16/uint16 [F,C,B] (box indices, e.g. ``argsort_volume``'s with
    ties rearranged as some machine's unstable ``scores.argsort()[::-1]`` left them, utils/nms.pyx:25), ncand int32 [F,C]
    (how many entries of each list are candidates).  Returns (keep_idx [F,C,cap], keep_cnt [F,C])."""
    if boxes.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    if order.dtype not in (torch.int16, torch.uint16) or ncand.dtype != torch.int32:
        raise ValueError("order must be (u)int16 [F,C,B], ncand int32 [F,C]")
    boxes, order, ncand = boxes.contiguous(), order.contiguous(), ncand.contiguous()
    F, B = boxes.shape[0], boxes.shape[1]
    C = order.shape[1]
    if tuple(order.shape) != (F, C, B) or tuple(ncand.shape) != (F, C) or boxes.shape[2] != 4:
        raise ValueError("boxes [F,B,4], order [F,C,B], ncand [F,C]")
    cap = B if cap is None else int(cap)
    ctx = _ctx_for(boxes, ctx)
    keep_idx = _keep_buffer((F, C, cap), boxes.device, pad)
    keep_cnt = torch.zeros((F, C), dtype=torch.int32, device=boxes.device)
    _finish(ctx, lambda: ctx.check(ctx.lib.vdet_nms_volume_ordered(
        ctx.h, boxes.data_ptr(), order.data_ptr(), ncand.data_ptr(), F, B, C, float(thresh), keep_idx.data_ptr(),
        keep_cnt.data_ptr(), cap)), True, reset=(lambda: keep_idx.fill_(-1)) if pad else None)
    return keep_idx, keep_cnt


def argsort_volume(scores, score_thresh=None, layout="FBC", ctx=None):
    """Descending argsort of every (frame, class) column of a score volume -- ``scores.argsort()[::-1]`` of
    utils/nms.pyx:25 / ``argsort(-cls_scores)`` of vdet/video_det.py:93 with the build's tie rule (equal scores by
    descending index, -0.0 == +0.0, NaN first).  Returns (order int16-as-uint16 [F,C,B] box indices, ncand int32 [F,C]);
    with score_thresh only boxes with score > score_thresh are candidates, the others form the tail."""
    if scores.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    scores = scores.contiguous()
    if layout == "FBC":
        F, B, C = scores.shape
        lay = _lib.LAYOUT_FBC
    elif layout == "FCB":
        F, C, B = scores.shape
        lay = _lib.LAYOUT_FCB
    else:
        raise ValueError("layout must be 'FBC' or 'FCB'")
    ctx = _ctx_for(scores, ctx)
    order = torch.empty((F, C, B), dtype=torch.int16, device=scores.device)     # uint16 payload (B <= 32767: never negative)
    ncand = torch.zeros((F, C), dtype=torch.int32, device=scores.device)
    _finish(ctx, lambda: ctx.check(ctx.lib.vdet_argsort_volume(
        ctx.h, scores.data_ptr(), lay, F, B, C, 0 if score_thresh is None else 1,
        0.0 if score_thresh is None else float(score_thresh), order.data_ptr(), ncand.data_ptr())), True)
    return order, ncand


def temporal_maxpool(vol, window, pad=-1e5, ctx=None):
    """Centred sliding max along axis 0 (array form of score_proto_temporal_maxpool,
    vdet/tubelet_cls.py:386-414; pad value :402).  vol: f32 [F, ...]."""
    if window % 2 != 1:
        raise ValueError('Window size must be odd!')
    if vol.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    vol = vol.contiguous()
    if window == 1:
        return vol.clone()          # (a copy like every other window size, never an alias of the input)
    out = torch.empty_like(vol)
    F = vol.shape[0]
    S = vol.numel() // F if F else 0
    ctx = _ctx_for(vol, ctx)
    ctx.check(ctx.lib.vdet_temporal_maxpool_f32(ctx.h, vol.data_ptr(), out.data_ptr(), F, S, int(window), float(pad)))
    return out


def temporal_conv(vol, taps, bias=0.0, pad=0.0, ctx=None):
    """Single-channel temporal convolution along axis 0 (build-defined stand-in for the external
    TCN of score_conv_cls, vdet/tubelet_cls.py:15-51): out[f] = bias + sum_k taps[k]*in[f+k-K/2]."""
    if vol.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    vol = vol.contiguous()
    t = np.ascontiguousarray(taps, dtype=np.float32)
    out = torch.empty_like(vol)
    F = vol.shape[0]
    S = vol.numel() // F if F else 0
    ctx = _ctx_for(vol, ctx)
    ctx.check(ctx.lib.vdet_temporal_conv_f32(ctx.h, vol.data_ptr(), out.data_ptr(), F, S, t.ctypes.data,
                                             t.shape[0], float(bias), float(pad)))
    return out


def temporal_maxpool_conv(vol, window, taps, pad_max=-1e5, bias=0.0, pad_conv=0.0, ctx=None):
    """``temporal_maxpool(vol, window, pad_max)`` and ``temporal_conv(vol, taps, bias, pad_conv)`` (len(taps) ==
    window) in one pass over the volume (it is read once; the pass is HBM-bound).  Returns (pooled, conv)."""
    if window % 2 != 1:
        raise ValueError('Window size must be odd!')
    if vol.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    t = np.ascontiguousarray(taps, dtype=np.float32)
    if t.shape[0] != window:
        raise ValueError('need one tap per window position')
    vol = vol.contiguous()
    out_m, out_c = torch.empty_like(vol), torch.empty_like(vol)
    F = vol.shape[0]
    S = vol.numel() // F if F else 0
    ctx = _ctx_for(vol, ctx)
    ctx.check(ctx.lib.vdet_temporal_maxpool_conv_f32(ctx.h, vol.data_ptr(), out_m.data_ptr(), out_c.data_ptr(), F, S,
                                                     int(window), float(pad_max), t.ctypes.data, float(bias), float(pad_conv)))
    return out_m, out_c


def volume_pass(scores, window=3, taps=None, pad_max=-1e5, bias=0.0, pad_conv=0.0, score_thresh=None, ctx=None, frame_off=None,
                out=None):
    """(``frame_off`` [V+1]: the volume is V videos concatenated along F -- a temporal window stops at its video's ends.)
    ONE read of a score volume [F,B,C]: ``temporal_maxpool(scores, window, pad_max)``, optionally
    ``temporal_conv(scores, taps, bias, pad_conv)`` (len(taps) == window), and -- left inside the context for
    the next ``nms_volume`` / ``track_volume`` / ``nms_track_volume`` call on the SAME scores tensor (cache
    enabled) -- the class-major sort keys of every (frame, class) problem (include/vdet_hip.h: vdet_volume_pass).
    Returns (pooled, conv or None).  ``out`` = (pooled buffer, conv buffer or None): caller-owned float32 tensors of at
    least ``scores.numel()`` elements to write into instead of fresh ones (a server keeps one pair per stream; the C-ABI
    takes caller buffers anyway); the returned tensors are views of them shaped like ``scores``."""
    if window % 2 != 1:
        raise ValueError('Window size must be odd!')
    if scores.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    if scores.dim() != 3:
        raise ValueError("scores must be [F,B,C]")
    t = None
    if taps is not None:
        t = np.ascontiguousarray(taps, dtype=np.float32)
        if t.shape[0] != window:
            raise ValueError('need one tap per window position')
    scores = scores.contiguous()
    F, B, C = scores.shape
    ctx = _ctx_for(scores, ctx)
    if out is None:
        out_m = torch.empty_like(scores)
        out_c = torch.empty_like(scores) if t is not None else None
    else:
        def view(buf):
            if buf.dtype != torch.float32 or not buf.is_contiguous() or buf.numel() < scores.numel() or buf.device != scores.device:
                raise ValueError("out buffers must be contiguous float32 tensors of >= scores.numel() elements on the scores' device")
            return buf.view(-1)[:scores.numel()].view(scores.shape)
        if len(out) != 2 or out[0] is None or (t is not None and out[1] is None):
            raise ValueError("out = (pooled buffer, conv buffer): the conv buffer is needed when taps are given")
        out_m = view(out[0])
        out_c = view(out[1]) if t is not None else None
        # the kernel reads ``scores`` while it streams both outputs: overlapping buffers would give silently wrong results
        spans = [(b.data_ptr(), b.data_ptr() + scores.numel() * 4) for b in (scores, out_m, out_c) if b is not None]
        for i in range(len(spans)):
            for j in range(i + 1, len(spans)):
                if spans[i][0] < spans[j][1] and spans[j][0] < spans[i][1]:
                    raise ValueError("out buffers must not overlap each other or scores")
    tail = (int(window), float(pad_max), t.ctypes.data if t is not None else None, float(bias), float(pad_conv),
            out_m.data_ptr(), out_c.data_ptr() if out_c is not None else None, 0 if score_thresh is None else 1,
            0.0 if score_thresh is None else float(score_thresh))
    if frame_off is None:
        ctx.check(ctx.lib.vdet_volume_pass(ctx.h, scores.data_ptr(), F, B, C, *tail))
    else:
        off = _frame_offsets(frame_off, F)
        ctx.check(ctx.lib.vdet_volume_pass_batch(ctx.h, scores.data_ptr(), off.ctypes.data, len(off) - 1, B, C, *tail))
    return out_m, out_c


def _frame_offsets(frame_off, F):
    off = np.ascontiguousarray(frame_off, dtype=np.int64).reshape(-1)
    if off.size < 2 or off[0] != 0 or off[-1] != F or np.any(np.diff(off) <= 0):
        raise ValueError("frame_off must run 0 = o[0] < o[1] < ... < o[V] = F")
    return off


def video_batch(boxes, scores, frame_off, nms_thres=0.3, thres=0.0, max_tracks=10, link_thres=0.5, max_frames=0, cap=None,
                nms=True, rescore=True, overlap_thres=0.7, window=3, sync=True, ctx=None, pad=True):
    """V small videos in ONE call (include/vdet_hip.h: vdet_video_batch): boxes [F,B,4] / scores [F,B,C] hold the frames
    of all videos one after the other, ``frame_off`` [V+1] their frame ranges.  Per video the results are what
    ``nms_track_volume`` + ``rescore_tracks`` return for it alone.  Returns a dict:
      keep_idx [F,C,cap] / keep_cnt [F,C] (nms), anchors [V,C,T,3], ntracks [V,C], and per-video VIEWS
      tracks[v] [C,T,F_v,5], det[v] / pooled[v] [C,T,F_v] f64, tboxes[v] [C,T,F_v,4] (rescore)."""
    if boxes.dtype != torch.float32 or scores.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    boxes, scores = boxes.contiguous(), scores.contiguous()
    F, B, C = scores.shape
    if tuple(boxes.shape) != (F, B, 4):
        raise ValueError("boxes must be [F,B,4]")
    off = _frame_offsets(frame_off, F)
    V, T = len(off) - 1, int(max_tracks)
    ctx = _ctx_for(boxes, ctx)
    dev = boxes.device
    cap = B if cap is None else int(cap)
    keep_idx = _keep_buffer((F, C, cap), dev, pad) if nms else None
    keep_cnt = torch.zeros((F, C), dtype=torch.int32, device=dev) if nms else None
    tracks = torch.full((C * max(T, 1) * F * 5,), float('nan'), dtype=torch.float32, device=dev)
    anchors = torch.zeros((V, C, max(T, 1), 3), dtype=torch.float32, device=dev)
    ntracks = torch.zeros((V, C), dtype=torch.int32, device=dev)
    det = torch.empty((C * max(T, 1) * F,), dtype=torch.float64, device=dev) if rescore else None
    pooled = torch.empty_like(det) if rescore else None
    tboxes = torch.empty((C * max(T, 1) * F * 4,), dtype=torch.float32, device=dev) if rescore else None

    def launch():
        ctx.check(ctx.lib.vdet_video_batch(
            ctx.h, boxes.data_ptr(), scores.data_ptr(), off.ctypes.data, V, B, C, float(nms_thres), float(thres), T, float(link_thres),
            int(max_frames), tracks.data_ptr(), anchors.data_ptr(), ntracks.data_ptr(), cap,
            keep_idx.data_ptr() if nms else None, keep_cnt.data_ptr() if nms else None, float(overlap_thres), int(window),
            det.data_ptr() if rescore else None, pooled.data_ptr() if rescore else None, tboxes.data_ptr() if rescore else None))

    def reset():
        tracks.fill_(float('nan'))
        if nms and pad:
            keep_idx.fill_(-1)

    _finish(ctx, launch, sync, reset=reset)
    out = dict(keep_idx=keep_idx, keep_cnt=keep_cnt, anchors=anchors[:, :, :T], ntracks=ntracks, frame_off=off)
    tv, dv, pv, bv = [], [], [], []
    for v in range(V):
        f0, fv = int(off[v]), int(off[v + 1] - off[v])
        tv.append(tracks[C * T * 5 * f0: C * T * 5 * (f0 + fv)].view(C, T, fv, 5))
        if rescore:
            dv.append(det[C * T * f0: C * T * (f0 + fv)].view(C, T, fv))
            pv.append(pooled[C * T * f0: C * T * (f0 + fv)].view(C, T, fv))
            bv.append(tboxes[C * T * 4 * f0: C * T * 4 * (f0 + fv)].view(C, T, fv, 4))
    out.update(tracks=tv, det=dv, pooled=pv, tboxes=bv)
    return out


def iou(boxes1, boxes2):
    """utils/common.py:451-468 -- float64 IoU matrix [n1,n2] (+1 convention), numpy in / numpy out."""
    b1 = np.ascontiguousarray(np.asarray(boxes1).astype('float'))
    b2 = np.ascontiguousarray(np.asarray(boxes2).astype('float'))
    if b1.ndim != 2 or b2.ndim != 2 or b1.shape[1] < 4 or b2.shape[1] < 4:
        raise IndexError("boxes must be [n,4]")
    b1 = np.ascontiguousarray(b1[:, :4])
    b2 = np.ascontiguousarray(b2[:, :4])
    out = np.empty((b1.shape[0], b2.shape[0]), dtype=np.float64)
    if out.size:
        ctx = _lib.get_context()
        ctx.reset_stream()
        ctx.check(ctx.lib.vdet_iou_f64(ctx.h, b1.ctypes.data, b1.shape[0], b2.ctypes.data, b2.shape[0],
                                       out.ctypes.data))
    return out


def track_volume(boxes, scores, nms_thres=0.3, thres=0.0, max_tracks=10, link_thres=0.5, max_frames=0, sync=True,
                 ctx=None):
    """Greedy tubelet generation for EVERY class of a video on the GPU: the array form of
    greedily_track_from_raw_dets (vdet/track.py:189-252) with the built-in IoU-linking tracker as
    ``track_method`` (the reference's trackers are external MATLAB code).

    boxes [F,B,4] f32, scores [F,B,C] f32.  Returns
      tracks  [C, max_tracks, F, 5] f32 rows (x1,y1,x2,y2,score), NaN where a track has no box,
      anchors [C, max_tracks, 3] f32 (1-based frame, box index, score),  ntracks [C] int32."""
    if boxes.dtype != torch.float32 or scores.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    boxes = boxes.contiguous()
    scores = scores.contiguous()
    F, B, C = scores.shape
    if tuple(boxes.shape) != (F, B, 4):
        raise ValueError("boxes must be [F,B,4]")
    ctx = _ctx_for(boxes, ctx)
    tracks = torch.full((C, max_tracks, F, 5), float('nan'), dtype=torch.float32, device=boxes.device)
    anchors = torch.zeros((C, max_tracks, 3), dtype=torch.float32, device=boxes.device)
    ntracks = torch.zeros((C,), dtype=torch.int32, device=boxes.device)
    _finish(ctx, lambda: ctx.check(ctx.lib.vdet_track_volume(
        ctx.h, boxes.data_ptr(), scores.data_ptr(), F, B, C, float(nms_thres), float(thres), int(max_tracks),
        float(link_thres), int(max_frames), tracks.data_ptr(), anchors.data_ptr(), ntracks.data_ptr())), sync,
        reset=lambda: (tracks.fill_(float('nan')), anchors.zero_()))
    return tracks, anchors, ntracks


def nms_track_volume(boxes, scores, nms_thres=0.3, thres=0.0, max_tracks=10, link_thres=0.5, max_frames=0, cap=None,
                     sync=True, ctx=None, pad=True, keep_out=None):
    """``nms_volume`` (layout 'FBC', no score threshold) and ``track_volume`` of the same video in one
    call: both are greedy walks over the same sorted lists and suppression graph, and on regular
    videos one fused walk serves both (include/vdet_hip.h: vdet_nms_track_volume).  Results are
    bit-identical to the two separate calls.  ``keep_out``: a caller-owned contiguous int32 tensor of >= F*C*cap elements
    to hold keep_idx (the returned keep_idx is a view of it).
    Returns (keep_idx, keep_cnt, tracks, anchors, ntracks)."""
    if boxes.dtype != torch.float32 or scores.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    boxes = boxes.contiguous()
    scores = scores.contiguous()
    F, B, C = scores.shape
    if tuple(boxes.shape) != (F, B, 4):
        raise ValueError("boxes must be [F,B,4]")
    cap = B if cap is None else int(cap)
    ctx = _ctx_for(boxes, ctx)
    if keep_out is None:
        keep_idx = _keep_buffer((F, C, cap), boxes.device, pad)
    else:
        if keep_out.dtype != torch.int32 or not keep_out.is_contiguous() or keep_out.numel() < F * C * cap or keep_out.device != boxes.device:
            raise ValueError("keep_out must be a contiguous int32 tensor of >= F*C*cap elements on the boxes' device")
        keep_idx = keep_out.view(-1)[:F * C * cap].view(F, C, cap)
        if pad:
            keep_idx.fill_(-1)
    keep_cnt = torch.zeros((F, C), dtype=torch.int32, device=boxes.device)
    tracks = torch.full((C, max_tracks, F, 5), float('nan'), dtype=torch.float32, device=boxes.device)
    anchors = torch.zeros((C, max_tracks, 3), dtype=torch.float32, device=boxes.device)
    ntracks = torch.zeros((C,), dtype=torch.int32, device=boxes.device)
    _finish(ctx, lambda: ctx.check(ctx.lib.vdet_nms_track_volume(
        ctx.h, boxes.data_ptr(), scores.data_ptr(), F, B, C, float(nms_thres), float(thres), int(max_tracks),
        float(link_thres), int(max_frames), tracks.data_ptr(), anchors.data_ptr(), ntracks.data_ptr(), cap,
        keep_idx.data_ptr(), keep_cnt.data_ptr())), sync,
        reset=lambda: (keep_idx.fill_(-1) if pad else None, tracks.fill_(float('nan')), anchors.zero_()))
    return keep_idx, keep_cnt, tracks, anchors, ntracks


def tracks_to_proto(video_name, tracks, anchors, ntracks, method='iou_link_tracker'):
    """One class's device tracks -> a .track protocol dict (utils/protocol.py:389-414 fields)."""
    from .utils.protocol import bbox_hash
    tr = tracks.cpu().numpy() if hasattr(tracks, 'cpu') else np.asarray(tracks)
    an = anchors.cpu().numpy() if hasattr(anchors, 'cpu') else np.asarray(anchors)
    out = []
    for t in range(int(ntracks)):
        anchor_frame = int(an[t, 0])
        tracklet = []
        for f in range(tr.shape[1]):
            row = tr[t, f]
            if np.isnan(row[0]):
                continue
            bbox = [int(v) for v in row[:4]]
            tracklet.append({'frame': f + 1, 'bbox': bbox, 'hash': bbox_hash(video_name, f + 1, row),
                             'score': float(row[4]), 'anchor': int(f + 1 - anchor_frame)})
        out.append(tracklet)
    return {'video': video_name, 'method': method, 'tracks': out}


def rescore_tracks(tracks, ntracks, boxes, scores, overlap_thres=0.7, window=3, sync=True, ctx=None):
    """Re-score device tracks: spatial max-pooling of the detection scores onto the tubelet boxes
    (raw_dets_spatial_max_pooling, vdet/tubelet_cls.py:493-535 -- also replaces each box by the
    best-scoring overlapping detection), gap completion (:284-303) and temporal max-pooling
    (:386-414).  Returns (det_score f64 [C,T,F], pooled f64 [C,T,F], boxes f32 [C,T,F,4])."""
    if window % 2 != 1:
        raise ValueError('Window size must be odd!')
    if tracks.dtype != torch.float32 or boxes.dtype != torch.float32 or scores.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    if ntracks.dtype != torch.int32:
        raise ValueError("ntracks must be int32")
    if tracks.dim() != 4 or tracks.shape[3] != 5:
        raise ValueError("tracks must be [C,T,F,5]")
    C, T, F = tracks.shape[0], tracks.shape[1], tracks.shape[2]
    if boxes.dim() != 3 or boxes.shape[0] != F or boxes.shape[2] != 4:
        raise ValueError("boxes must be [F,B,4]")
    B = boxes.shape[1]
    if tuple(scores.shape) != (F, B, C):
        raise ValueError("scores must be [F,B,C]")
    if tuple(ntracks.shape) != (C,):
        raise ValueError("ntracks must be [C]")
    for t in (tracks, ntracks, scores):
        if not t.is_cuda or t.device != boxes.device:
            raise ValueError("tracks, ntracks, boxes and scores must live on the same GPU")
    ntracks = ntracks.contiguous()
    ctx = _ctx_for(boxes, ctx)
    det = torch.empty((C, T, F), dtype=torch.float64, device=boxes.device)
    pooled = torch.empty((C, T, F), dtype=torch.float64, device=boxes.device)
    ob = torch.empty((C, T, F, 4), dtype=torch.float32, device=boxes.device)
    ctx.check(ctx.lib.vdet_rescore_tracks(ctx.h, tracks.contiguous().data_ptr(), ntracks.data_ptr(),
                                          boxes.contiguous().data_ptr(), scores.contiguous().data_ptr(), F, B, C, T,
                                          float(overlap_thres), int(window), det.data_ptr(), pooled.data_ptr(),
                                          ob.data_ptr()))
    if sync:
        ctx.sync()
    return det, pooled, ob
