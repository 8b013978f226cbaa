"""Per-frame scoring glue, video NMS wrapper and the Fast R-CNN per-class threshold / top-k
collection of the reference's vdet/video_det.py.  The CNN (``det_fun`` / ``net``) and the image
reader are external plug-ins; the selection and NMS run on the GPU."""
import copy
import os

import numpy as np

from .dataset import imagenet_vdet_classes
from ..utils.protocol import empty_det_from_box, score_proto, det_score, boxes_at_frame, frame_path_at
from ..utils.common import imread
from ..utils.log import logger as logging
from ..utils.timer import Timer
from ..utils.cython_nms import vid_nms
from .. import hot


def det_vid_with_box(vid_proto, box_proto, det_fun, net, class_names=imagenet_vdet_classes):
    """:14-31 -- scores every proposal of every frame with ``det_fun(img, boxes, net)``."""
    assert vid_proto['video'] == box_proto['video']
    root = vid_proto['root_path']
    det_proto = empty_det_from_box(box_proto)
    for frame in vid_proto['frames']:
        frame_id, path = frame['frame'], frame['path']
        det_cur_frame = [i for i in det_proto['detections'] if i['frame'] == frame_id]
        if len(det_cur_frame) > 0:
            logging.info("Detecting in frame {}, {} boxes...".format(frame_id, len(det_cur_frame)))
            img = imread(os.path.join(root, path))
            boxes = [det['bbox'] for det in det_cur_frame]
            det_scores = det_fun(img, boxes, net)
            for det, scores in zip(det_cur_frame, det_scores):
                det['scores'] = score_proto(class_names, scores)
    return det_proto


def det_vid_without_box(vid_proto, det_fun, net, class_names=imagenet_vdet_classes):
    """:34-40 -- proposals come from the external MATLAB selective search (vdet/proposal.py),
    which is outside this build: supply a box_proto instead."""
    raise RuntimeError("region proposals are an external engine (MATLAB selective search, "
                       "reference vdet/proposal.py); call det_vid_score with a box_proto")


def det_vid_score(vid_proto, det_fun, net, box_proto=None, class_names=imagenet_vdet_classes):
    if box_proto:
        return det_vid_with_box(vid_proto, box_proto, det_fun, net, class_names)
    return det_vid_without_box(vid_proto, det_fun, net, class_names)


def apply_vid_nms(det_proto, class_index, thres=0.3):
    """:51-61 -- NOTE the reference ignores ``thres`` and always uses 0.3 (:57); kept."""
    logging.info('Apply NMS on video: {}'.format(det_proto['video']))
    boxes = np.asarray([[det['frame'], ] + det['bbox'] + [det_score(det, class_index), ]
                        for det in det_proto['detections']], dtype='float32')
    if boxes.ndim != 2:
        boxes = boxes.reshape(0, 6)
    keep = vid_nms(boxes, thresh=0.3)
    new_det = {'video': det_proto['video'],
               'detections': [det_proto['detections'][i] for i in keep]}
    logging.info("{} / {} windows kept.".format(len(new_det['detections']), len(det_proto['detections'])))
    return new_det


def fast_rcnn_det_vid(net, vid_proto, box_proto, det_fun, class_names=imagenet_vdet_classes,
                      max_per_image=100, thresh=0.05):
    """:64-106 -- all_boxes[cls][frame] = float32 [n,5] (x1,y1,x2,y2,score) of the boxes with
    score > thresh, cut to the max_per_image best.  ``det_fun(net, im, boxes) -> (scores
    [B,C+1], boxes [B,4(C+1)])`` is the external CNN."""
    num_images = len(vid_proto['frames'])
    num_classes = len(class_names)
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    _t = {'im_detect': Timer(), 'misc': Timer()}
    for i, frame in enumerate(vid_proto['frames']):
        im = imread(frame_path_at(vid_proto, frame['frame']))
        _t['im_detect'].tic()
        orig_boxes = np.array([box['bbox'] for box in boxes_at_frame(box_proto, frame['frame'])])
        scores, boxes = det_fun(net, im, orig_boxes)
        _t['im_detect'].toc()

        _t['misc'].tic()
        scores = np.asarray(scores)
        boxes = np.asarray(boxes)
        selected = hot.threshold_topk(scores[:, :num_classes], thresh, max_per_image, col0=1)
        for j in range(1, num_classes):
            inds = selected[j - 1]
            all_boxes[j][i] = np.hstack((boxes[inds, j * 4:(j + 1) * 4], scores[inds, j][:, np.newaxis])) \
                .astype(np.float32, copy=False)
        _t['misc'].toc()
        logging.info('im_detect: {:d}/{:d} {:.3f}s {:.3f}s'.format(
            i + 1, num_images, _t['im_detect'].average_time, _t['misc'].average_time))
    return all_boxes
