"""Still-image NMS wrapper of the reference's vdet/image_det.py (:117-123).  The CNN scorers of
that file (:12-114: Fast R-CNN / GoogLeNet R-CNN forward passes, SVM scoring) are external engines
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
