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


def _ctx_for(t):
    if not t.is_cuda:
        raise ValueError("expected a CUDA/HIP tensor (vdetlib_amd has no CPU path)")
    ctx = _lib.get_context(t.device.index if t.device.index is not None else torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
    return ctx


def nms_volume(boxes, scores, thresh=0.3, score_thresh=None, cap=None, layout="FBC", sync=True):
    """Per-(frame, class) greedy NMS of a whole video (vdet/image_det.py:117-123 applied to every
    frame and class of vdet/video_det.py:89-99; == utils/nms.pyx vid_nms per class).

    boxes [F,B,4] f32, scores [F,B,C] (layout 'FBC', class innermost like zs[B,C]) or [F,C,B] ('FCB').
    Returns (keep_idx int32 [F,C,cap], keep_cnt int32 [F,C]); keep_idx[f,c,:cnt] are box indices in
    descending score order, the rest is -1.
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
    ctx = _ctx_for(boxes)
    keep_idx = torch.full((F, C, cap), -1, dtype=torch.int32, device=boxes.device)
    keep_cnt = torch.zeros((F, C), dtype=torch.int32, device=boxes.device)
    ctx.check(ctx.lib.vdet_nms_volume(ctx.h, boxes.data_ptr(), scores.data_ptr(), lay, F, B, C, float(thresh),
                                      0 if score_thresh is None else 1,
                                      0.0 if score_thresh is None else float(score_thresh),
                                      keep_idx.data_ptr(), keep_cnt.data_ptr(), cap))
    if sync:
        ctx.sync()
    return keep_idx, keep_cnt


def temporal_maxpool(vol, window, pad=-1e5):
    """Centred sliding max along axis 0 (array form of score_proto_temporal_maxpool,
    vdet/tubelet_cls.py:386-414; pad value :402).  vol: f32 [F, ...]."""
    if window % 2 != 1:
        raise ValueError('Window size must be odd!')
    if vol.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    vol = vol.contiguous()
    if window == 1:
        return vol
    out = torch.empty_like(vol)
    F = vol.shape[0]
    S = vol.numel() // F if F else 0
    ctx = _ctx_for(vol)
    ctx.check(ctx.lib.vdet_temporal_maxpool_f32(ctx.h, vol.data_ptr(), out.data_ptr(), F, S, int(window), float(pad)))
    return out


def temporal_conv(vol, taps, bias=0.0, pad=0.0):
    """Single-channel temporal convolution along axis 0 (build-defined stand-in for the external
    TCN of score_conv_cls, vdet/tubelet_cls.py:15-51): out[f] = bias + sum_k taps[k]*in[f+k-K/2]."""
    if vol.dtype != torch.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t'")
    vol = vol.contiguous()
    t = np.ascontiguousarray(taps, dtype=np.float32)
    out = torch.empty_like(vol)
    F = vol.shape[0]
    S = vol.numel() // F if F else 0
    ctx = _ctx_for(vol)
    ctx.check(ctx.lib.vdet_temporal_conv_f32(ctx.h, vol.data_ptr(), out.data_ptr(), F, S, t.ctypes.data,
                                             t.shape[0], float(bias), float(pad)))
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
