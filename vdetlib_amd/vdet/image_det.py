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
    """reference vdet/image_det.py:109-114: ``features * (20 / feat_norm_mean) @ W + B``, the path's one
    dense contraction ([n, 1024] x [1024, 200]).  A plain library GEMM: the product
    runs on the GPU through torch.matmul (rocBLAS / hipBLASLt) in numpy's result dtype (float64 for the
    reference's .mat models), everything around it is numpy as in the reference.  No CPU fallback."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("vdetlib_amd.vdet.image_det.svm_scores needs a HIP device (no CPU fallback)")
    features = np.asarray(features)
    if features.ndim == 4:
        features = np.squeeze(features, axis=(2, 3))
    features = np.asarray(features) * (20. / svm_model['feat_norm_mean'])
    W = np.asarray(svm_model['W'])
    dt = np.result_type(features.dtype, W.dtype)
    prod = torch.matmul(torch.from_numpy(np.ascontiguousarray(features, dtype=dt)).cuda(),
                        torch.from_numpy(np.ascontiguousarray(W, dtype=dt)).cuda()).cpu().numpy()
    return prod + svm_model['B']
