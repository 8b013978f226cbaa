"""Still-image NMS wrapper of the reference's vdet/image_det.py (:117-123).  The CNN scorers of
that file (:12-106: Fast R-CNN / GoogLeNet R-CNN forward passes) are external engines
and out of scope."""
import numpy as np

from ..utils.cython_nms import nms
from ..utils.log import logger as logging


def apply_image_nms(boxes, scores, thres=0.3):
    """boxes [N,4], scores [N] (any float dtype) -> keep list, descending score (:117-123)."""
    box_score = np.asarray(np.r_['-1', boxes, np.reshape(scores, (-1, 1))], dtype='float32')
    logging.info("Applying nms to image.")
    keep = nms(box_score, thres)
    logging.info("{} / {} boxes kept.".format(len(keep), len(boxes)))
    return keep


def svm_scores(features, svm_model):
    """reference vdet/image_det.py:109-114: ``features * (20 / feat_norm_mean) @ W + B``, the path's one dense
    contraction ([n, 1024] x [1024, 200]), on the GPU as a hand-written MFMA kernel
    (``vdet_svm_scores_f64`` / ``_f32``: v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32) in numpy's result
    dtype -- float64 for the reference's .mat models, float32 when features and W both are.  The scaling, the
    squeeze of [n,k,1,1] features and the broadcast of B are the reference's; no CPU fallback."""
    from .. import _lib
    features = np.asarray(features)
    if features.ndim == 4:
        features = np.squeeze(features, axis=(2, 3))
    features = np.asarray(features) * (20. / svm_model['feat_norm_mean'])
    W = np.asarray(svm_model['W'])
    B = np.asarray(svm_model['B'])
    if features.ndim != 2 or W.ndim != 2 or features.shape[1] != W.shape[0]:
        raise ValueError("shapes %s and %s not aligned" % (features.shape, W.shape))
    # the product is computed in the dtype numpy's dot would use (features, W); B joins afterwards exactly as
    # ``+ B`` would (a float64 B on float32 operands promotes the SUM, not the contraction)
    dt = np.result_type(features.dtype, W.dtype)
    dt = np.dtype(np.float32) if dt == np.float32 else np.dtype(np.float64)
    a = np.ascontiguousarray(features, dtype=dt)
    w = np.ascontiguousarray(W, dtype=dt)
    n, k = a.shape
    m = w.shape[1]
    # the bias is fused only when it is a per-COLUMN row ([m] / [1, m]) of the product's dtype; anything else -- a
    # column-shaped [m, 1] B (numpy broadcasts it per row when n == m and raises otherwise), another dtype --
    # is added by numpy with numpy's own broadcasting rules
    fuse = (B.ndim <= 1 or B.shape == (1, m)) and B.size == m and np.result_type(dt, B.dtype) == dt
    bias = np.ascontiguousarray(B.reshape(m), dtype=dt) if fuse else None
    out = np.empty((n, m), dtype=dt)
    # a private context on its own stream: the shared per-device context (and whichever torch stream it follows)
    # is left alone
    global _svm_ctx
    if _svm_ctx is None:
        _svm_ctx = _lib.Context()
    ctx = _svm_ctx
    fn = ctx.lib.vdet_svm_scores_f32 if dt == np.float32 else ctx.lib.vdet_svm_scores_f64
    ctx.check(fn(ctx.h, a.ctypes.data, n, k, w.ctypes.data, bias.ctypes.data if bias is not None else None, m,
                 out.ctypes.data))
    return out if bias is not None else out + B


_svm_ctx = None
