"""Video-level detection stage (the API of the reference's vdet/video_det.py: same function names, argument
order, defaults, dict keys and return containers), organised around ARRAYS instead of per-detection python
lists: detections are grouped by frame once, the rows handed to the native NMS are assembled column-wise,
and the per-class candidate selection of every frame runs on the GPU (``hot.threshold_topk`` ->
``vdet_threshold_topk``; for device-resident score volumes the whole threshold -> top-k -> NMS chain is
``ops.nms_volume(..., score_thresh=, topk=)``).  The CNN (``det_fun`` / ``net``) and the image reader are
external plug-ins, exactly as in the reference."""
import operator
import os
from collections import defaultdict
from itertools import chain, islice, repeat
from operator import itemgetter

import numpy as np

from .dataset import imagenet_vdet_classes
from ..utils.protocol import empty_det_from_box, score_proto, boxes_at_frame, frame_path_at
from ..utils.common import imread
from ..utils.log import logger as logging
from ..utils.timer import Timer
from ..utils.cython_nms import vid_nms
from .. import hot

try:                      # host-side marshalling helper (plain C, built by csrc/build.sh); the itertools form below is the same loop
    from .. import _protofast
except ImportError:       # pragma: no cover
    _protofast = None


def _by_frame(items):
    """{frame id: [items of that frame, in their original order]} -- one pass, instead of a rescan per frame."""
    groups = defaultdict(list)
    for it in items:
        groups[it['frame']].append(it)
    return groups


def det_vid_with_box(vid_proto, box_proto, det_fun, net, class_names=imagenet_vdet_classes):
    """Score every proposal with ``det_fun(img, boxes, net)`` (reference vdet/video_det.py:14-31).  Like the
    reference, the returned det_proto REUSES the box dicts of ``box_proto`` (each gains a 'scores' list),
    and frames without proposals are not read."""
    assert vid_proto['video'] == box_proto['video']
    det_proto = empty_det_from_box(box_proto)
    per_frame = _by_frame(det_proto['detections'])
    for frame in vid_proto['frames']:
        dets = per_frame.get(frame['frame'])
        if not dets:
            continue
        logging.info("Detecting in frame {}, {} boxes...".format(frame['frame'], len(dets)))
        img = imread(os.path.join(vid_proto['root_path'], frame['path']))
        for det, scores in zip(dets, det_fun(img, [d['bbox'] for d in dets], net)):
            det['scores'] = score_proto(class_names, scores)
    return det_proto


def det_vid_without_box(vid_proto, det_fun, net, class_names=imagenet_vdet_classes):
    """Reference :34-40 generates the proposals with MATLAB selective search (vdet/proposal.py), an external
    engine outside this build: supply a box_proto."""
    raise RuntimeError("region proposals are an external engine (MATLAB selective search, "
                       "reference vdet/proposal.py); call det_vid_score with a box_proto")


def det_vid_score(vid_proto, det_fun, net, box_proto=None, class_names=imagenet_vdet_classes):
    """Reference :43-48."""
    scorer = det_vid_with_box if box_proto else det_vid_without_box
    args = (vid_proto, box_proto, det_fun, net) if box_proto else (vid_proto, det_fun, net)
    return scorer(*args, class_names=class_names)


def _vid_nms_rows(detections, class_index):
    """float32 [N,6] rows (frame, x1,y1,x2,y2, score of ``class_index``) of a det_proto, assembled by column with C-level
    iteration (itemgetter / chain: no python-level loop over the detections).  The class score is ``det_score``'s
    (utils/protocol.py:323-327): the FIRST entry whose 'class_index' equals ``class_index``, -inf when there is none."""
    n = len(detections)
    rows = np.empty((n, 6), dtype=np.float32)
    if n and _protofast is not None:     # the same dict reads through the C API (csrc/protofast.c): same keys, same first-match rule
        _protofast.vid_nms_rows(detections, class_index, rows)
        return rows
    if n:
        rows[:, 0] = np.fromiter(map(_FRAME, detections), dtype=np.float64, count=n)
        try:
            rows[:, 1:5] = np.fromiter(chain.from_iterable(map(_BBOX, detections)), dtype=np.float64, count=4 * n).reshape(n, 4)
        except (ValueError, TypeError):      # (a bbox that is not four plain numbers: numpy's own conversion and errors)
            rows[:, 1:5] = np.asarray([d['bbox'] for d in detections], dtype=np.float64).reshape(n, 4)
        entries = _positional_entries(detections, class_index)
        if entries is not None:
            rows[:, 5] = np.fromiter(map(_SCORE, entries), dtype=np.float64, count=n)
        else:
            rows[:, 5] = np.fromiter((_class_score(d['scores'], class_index) for d in detections), dtype=np.float64, count=n)
    return rows


_FRAME, _BBOX, _SCORES, _SCORE, _CLASS_INDEX = (itemgetter(k) for k in ('frame', 'bbox', 'scores', 'score', 'class_index'))


def _positional_entries(detections, class_index):
    """``score_proto`` (utils/protocol.py:307-320) lists the classes in index order, so a detection's entry for class k normally
    sits AT position k.  Returns those entries when that is ``det_score``'s answer for EVERY detection -- the entry at the
    position has the class index and no EARLIER entry has it too (first match wins in the reference) -- else None (the caller
    then scans per detection like the reference)."""
    try:
        k = operator.index(class_index)
    except TypeError:
        return None
    if k < 0:
        return None
    try:
        lists = list(map(_SCORES, detections))
        entries = list(map(itemgetter(k), lists))
        if not all(map(operator.eq, map(_CLASS_INDEX, entries), repeat(class_index))):
            return None
        if k and class_index in map(_CLASS_INDEX, chain.from_iterable(map(islice, lists, repeat(k)))):
            return None
    except (IndexError, TypeError, KeyError):
        return None
    return entries


def _class_score(scores, class_index):
    """``det_score`` (utils/protocol.py:323-327): the score whose 'class_index' equals ``class_index`` (first match), -inf when
    the detection has none."""
    for e in scores:
        if e['class_index'] == class_index:
            return e['score']
    return float('-inf')


def apply_vid_nms(det_proto, class_index, thres=0.3):
    """Per-class video NMS (reference :51-61): detections of different frames never suppress each other; the
    survivors come back in descending score order.  NOTE: the reference ignores ``thres`` and always
    suppresses at 0.3 (:57) -- kept, callers get the reference's results."""
    logging.info('Apply NMS on video: {}'.format(det_proto['video']))
    dets = det_proto['detections']
    keep = vid_nms(_vid_nms_rows(dets, class_index), thresh=0.3)
    kept = {'video': det_proto['video'], 'detections': [dets[i] for i in keep]}
    logging.info("{} / {} windows kept.".format(len(kept['detections']), len(dets)))
    return kept


def fast_rcnn_det_vid(net, vid_proto, box_proto, det_fun, class_names=imagenet_vdet_classes,
                      max_per_image=100, thresh=0.05):
    """Fast R-CNN over a video (reference :64-106): ``det_fun(net, im, boxes) -> (scores [B,C+1],
    boxes [B,4(C+1)])`` is the external CNN; returns all_boxes[cls][frame] = float32 [n,5]
    (x1,y1,x2,y2,score) of the proposals scoring > thresh, cut to the max_per_image best (class 0 =
    background stays empty).  The selection of all classes of a frame is one GPU call."""
    frames = vid_proto['frames']
    num_classes = len(class_names)
    all_boxes = [[[] for _ in frames] for _ in range(num_classes)]
    timers = {'im_detect': Timer(), 'misc': Timer()}
    for i, frame in enumerate(frames):
        im = imread(frame_path_at(vid_proto, frame['frame']))
        timers['im_detect'].tic()
        proposals = np.array([box['bbox'] for box in boxes_at_frame(box_proto, frame['frame'])])
        scores, boxes = (np.asarray(a) for a in det_fun(net, im, proposals))
        timers['im_detect'].toc()

        timers['misc'].tic()
        # per class j >= 1: indices of the boxes with score > thresh, best max_per_image first when cut
        picked = hot.threshold_topk(scores[:, :num_classes], thresh, max_per_image, col0=1)
        for j, inds in enumerate(picked, start=1):
            all_boxes[j][i] = np.concatenate((boxes[inds, 4 * j:4 * j + 4], scores[inds, j, None]), axis=1) \
                .astype(np.float32, copy=False)
        timers['misc'].toc()
        logging.info('im_detect: {:d}/{:d} {:.3f}s {:.3f}s'.format(
            i + 1, len(frames), timers['im_detect'].average_time, timers['misc'].average_time))
    return all_boxes
