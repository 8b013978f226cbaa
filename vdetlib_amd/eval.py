"""VOC-style average precision for video object detection (the reference has no
evaluator; BASELINE config 5 asks for mAP on VID-shaped data).  Host-side tool, not on the hot path.

Detections are scored boxes per (video, frame, class); ground truth comes from .annot protocol
dicts (tools/imagenet_annotation_processor).  A detection is a true positive when it overlaps a
not-yet-matched ground-truth box of its video/frame/class with IoU >= iou_thr (+1 pixel
convention, the reference's utils/common.py:451-468); AP is the area under the monotone
precision envelope (VOC2010+ / ILSVRC all-point interpolation)."""
from collections import defaultdict

import numpy as np


def _iou_1n(box, boxes):
    ix1 = np.maximum(box[0], boxes[:, 0]); iy1 = np.maximum(box[1], boxes[:, 1])
    ix2 = np.minimum(box[2], boxes[:, 2]); iy2 = np.minimum(box[3], boxes[:, 3])
    iw = np.maximum(0.0, ix2 - ix1 + 1); ih = np.maximum(0.0, iy2 - iy1 + 1)
    inter = iw * ih
    a = (box[2] - box[0] + 1) * (box[3] - box[1] + 1)
    b = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    return inter / (a + b - inter)


def ground_truth_from_annots(annot_protos):
    """{(video, frame, class_index): float64 [n,4]}"""
    gt = defaultdict(list)
    for annot in annot_protos:
        for track in annot['annotations']:
            for box in track['track']:
                gt[(annot['video'], box['frame'], box['class_index'])].append(box['bbox'])
    return {k: np.asarray(v, dtype=np.float64).reshape(-1, 4) for k, v in gt.items()}


def detections_from_score_protos(score_protos, key='det_score'):
    """Tubelet boxes of .score protocol dicts -> list of (video, frame, class_index, bbox, score)."""
    dets = []
    for sp in score_protos:
        for tubelet in sp['tubelets']:
            for box in tubelet['boxes']:
                dets.append((sp['video'], box['frame'], tubelet['class_index'], box['bbox'], float(box[key])))
    return dets


def detections_from_tracks(video, tracks, ntracks, scores, boxes=None):
    """Device tubelets (ops.track_volume / ops.rescore_tracks arrays, already on the host) -> the same
    list.  tracks [C,T,F,5]; scores [C,T,F] (NaN = no box); boxes [C,T,F,4] (default: the track boxes).
    class_index = c + 1 (column c of the score volume is class c + 1, vdet/tubelet_cls.py:514)."""
    dets = []
    C, T, F = scores.shape
    bx = tracks[..., :4] if boxes is None else boxes
    for c in range(C):
        for t in range(int(ntracks[c])):
            for f in range(F):
                if not np.isnan(scores[c, t, f]):
                    dets.append((video, f + 1, c + 1, [float(v) for v in bx[c, t, f]], float(scores[c, t, f])))
    return dets


def average_precision(tp, n_gt):
    """tp: bool array in descending-score order."""
    if n_gt == 0:
        return float('nan')
    tp = np.asarray(tp, dtype=bool)
    ctp = np.cumsum(tp); cfp = np.cumsum(~tp)
    rec = ctp / float(n_gt)
    prec = ctp / np.maximum(ctp + cfp, 1)
    mrec = np.concatenate([[0.0], rec, [1.0]])
    mpre = np.concatenate([[0.0], prec, [0.0]])
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def evaluate(dets, gt, iou_thr=0.5, classes=None):
    """dets: list of (video, frame, class_index, bbox, score); gt from ground_truth_from_annots.
    Returns ({class_index: AP}, mAP over the classes that have ground truth)."""
    by_class = defaultdict(list)
    for d in dets:
        by_class[d[2]].append(d)
    gt_classes = sorted(set(k[2] for k in gt)) if classes is None else list(classes)
    aps = {}
    for c in gt_classes:
        n_gt = sum(len(v) for k, v in gt.items() if k[2] == c)
        cd = sorted(by_class.get(c, []), key=lambda d: -d[4])      # stable: ties keep input order
        matched = {}
        tp = np.zeros(len(cd), dtype=bool)
        for i, (video, frame, _, bbox, _) in enumerate(cd):
            g = gt.get((video, frame, c))
            if g is None or len(g) == 0:
                continue
            ious = _iou_1n(np.asarray(bbox, dtype=np.float64), g)
            used = matched.setdefault((video, frame), np.zeros(len(g), dtype=bool))
            ious = np.where(used, -1.0, ious)
            j = int(np.argmax(ious))
            if ious[j] >= iou_thr:
                tp[i] = True
                used[j] = True
        aps[c] = average_precision(tp, n_gt)
    valid = [v for v in aps.values() if not np.isnan(v)]
    return aps, (float(np.mean(valid)) if valid else float('nan'))
