"""Tubelet re-scoring of the reference's vdet/tubelet_cls.py: spatial max-pooling of detection
scores onto tubelet boxes, gap completion, temporal max-pooling, interpolation, and the channel
assembly of the temporal-convolution scorer.  Dict plumbing in python, every numeric core on the
GPU (vdetlib_amd.hot -> include/vdet_hip.h).

Out of scope (external Caffe/SVM engines): fast_rcnn_cls, googlenet_cls,
rcnn_scoring, rcnn_sampling_scoring, rcnn_sampling_dets_scoring, sampling_boxes (:53-260).
"""
import copy
from collections import defaultdict

import numpy as np

from ..utils.protocol import tubelets_overlap, tubelets_proto_from_tracks_proto
from ..utils.common import iou
from ..utils.log import logger as logging
from .dataset import imagenet_vdet_classes
from .. import hot


# per-box fields that become channel sequences of the temporal-convolution scorer: blob name -> box key
_TCN_BOX_FIELDS = (('det_scores', 'det_score'), ('track_scores', 'track_score'), ('gt_overlaps', 'gt_overlap'))
_TCN_OPTIONAL_FIELDS = (('all_scores', 'all_score'), ('feats', 'feat'))       # memory-heavy: read only when the net has the blob


def _tcn_channels(tubelet, blob_names):
    """Everything the reference offers a net per tubelet (:19-37), keyed by blob name: the per-box fields, the anchor offset
    normalised by the tubelet length (float division) and its magnitude, the 0 / 1 labels (gt overlap >= 0.5), and the
    tubelet-level entries `length`, `gt`, `mean_iou` (a net with blobs of those names receives them too)."""
    boxes = tubelet['boxes']
    length = len(boxes)
    ch = {name: [box[key] for box in boxes] for name, key in _TCN_BOX_FIELDS}
    rel = np.asarray([box['anchor'] for box in boxes], dtype=np.float64) / length
    ch['anchors'] = rel
    ch['abs_anchors'] = np.abs(rel)
    ch['labels'] = (np.asarray(ch['gt_overlaps'], dtype=np.float64) >= 0.5).astype(np.int64)
    ch['length'], ch['gt'] = length, tubelet['gt']
    ch['mean_iou'] = np.mean([ch['gt_overlaps']])
    for name, key in _TCN_OPTIONAL_FIELDS:
        if name in blob_names:
            ch[name] = [box[key] for box in boxes]
    return ch, length


def score_conv_cls(score_proto, net):
    """Reference :15-51.  Contract: for every tubelet, each blob of ``net`` whose name is one of the channel names above is
    reshaped to (1, its channel count, 1, L) and filled with the float32 sequence; ``net.forward()['probs'][:, 1, :]``
    becomes ``box['conv_score']`` (python float) of the tubelet's boxes, in place; the returned proto is a shallow copy.
    ``net`` is any object with ``.blobs`` (name -> object with ``.shape``, ``.reshape(*dims)``, ``.data``) and
    ``.forward()`` -- pycaffe's interface; ``vdetlib_amd.vdet.tcn.TCNNet`` is a gfx950 implementation of it."""
    out = copy.copy(score_proto)
    print("{}: {} tubelet(s).".format(score_proto['video'], len(out['tubelets'])))
    blob_names = set(net.blobs.keys())
    for tubelet in out['tubelets']:
        channels, length = _tcn_channels(tubelet, blob_names)
        for name in blob_names.intersection(channels):
            blob = net.blobs[name]
            blob.reshape(1, blob.shape[1], 1, length)
            blob.data[...] = np.asarray(channels[name], dtype='float32')
        probs = np.asarray(net.forward()['probs'][:, 1, :]).ravel()
        for box, p in zip(tubelet['boxes'], probs):
            box['conv_score'] = float(p)
    return out


def scoring_tracks(vid_proto, track_proto, annot_proto, sc_method, net, class_idx):
    """:263-273"""
    assert vid_proto['video'] == track_proto['video']
    tubelets_proto = sc_method(vid_proto, track_proto, net, class_idx)
    if annot_proto is not None:
        tubelets_proto = tubelets_overlap(tubelets_proto, annot_proto, class_idx)
    return {'video': vid_proto['video'], 'method': sc_method.__name__, 'tubelets': tubelets_proto}


def classify_tracks(video_proto, track_proto, cls_method, net, class_idx):
    """:276-282"""
    assert video_proto['video'] == track_proto['video']
    return {'video': video_proto['video'], 'method': cls_method.__name__,
            'tracks': cls_method(video_proto, track_proto, net, class_idx)}


def do_score_completion(score_proto):
    """:284-303 -- in place: runs of det_score <= -10 are filled (edge extension / linear
    interpolation).  IndexError, like the reference, for a tubelet with no valid score at all."""
    tubelets = score_proto['tubelets']
    series = [[box['det_score'] for box in tubelet['boxes']] for tubelet in tubelets]
    done = hot.series_completion(series)
    for tubelet, new in zip(tubelets, done):
        for box, old, v in zip(tubelet['boxes'], [b['det_score'] for b in tubelet['boxes']], new):
            if old <= -10:            # untouched entries keep their python object (int stays int)
                box['det_score'] = float(v)


def _spatial_max_pooling(vid_proto, tubelets_proto, frame_dets, overlap_thres):
    """Shared core of :316-347 / :504-532.  frame_dets: frame_id -> (det_boxes array, det_scores
    1-D array) for the frames that have detections."""
    frame_to_tubelets_idx = defaultdict(list)
    for i, tubelet in enumerate(tubelets_proto):
        for j, box in enumerate(tubelet['boxes']):
            frame_to_tubelets_idx[box['frame']].append((i, j))
    slots, det_boxes_list, det_scores_list = {}, [], []
    tub_boxes, tub_group, targets = [], [], []
    for frame in vid_proto['frames']:
        frame_id = frame['frame']
        if frame_id not in frame_dets:
            continue
        for i, j in frame_to_tubelets_idx[frame_id]:
            cur_box = tubelets_proto[i]['boxes'][j]
            if cur_box is None:
                continue
            if frame_id not in slots:
                slots[frame_id] = len(det_boxes_list)
                det_boxes_list.append(frame_dets[frame_id][0])
                det_scores_list.append(frame_dets[frame_id][1])
            tub_boxes.append(cur_box['bbox'])
            tub_group.append(slots[frame_id])
            targets.append((i, j, frame_id))
    idx, score = hot.spatial_maxpool(tub_boxes, tub_group, det_boxes_list, det_scores_list, overlap_thres)
    for (i, j, frame_id), k, s in zip(targets, idx, score):
        cur_box = tubelets_proto[i]['boxes'][j]
        if k >= 0:
            cur_box['det_score'] = float(s)
            cur_box['bbox'] = frame_dets[frame_id][0][int(k)].tolist()
        else:
            print("Warning: Tubelet {} has no overlapping dets (IOU > {}).".format(i, overlap_thres))
            cur_box['det_score'] = float(-1e5)


def dets_spatial_max_pooling(vid_proto, track_proto, det_proto, class_idx, overlap_thres=0.7):
    """:305-350 -- detections as protocol dicts; the class score is read BY POSITION
    ``det['scores'][class_idx - 1]['score']`` (:329)."""
    assert vid_proto['video'] == track_proto['video']
    score_proto = {'video': vid_proto['video'],
                   'method': "spatial_max_pooling_IOU_{}".format(overlap_thres)}
    tubelets_proto = tubelets_proto_from_tracks_proto(track_proto['tracks'], class_idx)
    logging.info("Sampling dets in {} for {}...".format(vid_proto['video'], imagenet_vdet_classes[class_idx]))
    frame_to_det_idx = defaultdict(list)
    for i, det in enumerate(det_proto['detections']):
        frame_to_det_idx[det['frame']].append(i)
    frame_dets = {}
    for frame_id, det_idx in frame_to_det_idx.items():
        frame_dets[frame_id] = (
            np.asarray([det_proto['detections'][i]['bbox'] for i in det_idx]),
            np.asarray([det_proto['detections'][i]['scores'][class_idx - 1]['score'] for i in det_idx]))
    _spatial_max_pooling(vid_proto, tubelets_proto, frame_dets, overlap_thres)
    score_proto['tubelets'] = tubelets_proto
    do_score_completion(score_proto)
    return score_proto


class _ClassDetsByFrame(object):
    """frame -> (bboxes [n,4], class scores [n]) of a det_proto, detections in protocol order.  Only the frames that are
    asked for are touched (and remembered): a det_proto whose detections of OTHER frames carry short score lists must not
    fail here -- the reference only reads the anchor frames' detections (:366-374)."""

    def __init__(self, det_proto, class_idx):
        self._dets = det_proto['detections']
        self._cls = class_idx - 1
        self._by_frame = defaultdict(list)
        for i, det in enumerate(self._dets):
            self._by_frame[det['frame']].append(i)
        self._memo = {}

    def __call__(self, frame):
        got = self._memo.get(frame)
        if got is None:
            idx = self._by_frame.get(frame, ())
            got = self._memo[frame] = (np.asarray([self._dets[i]['bbox'] for i in idx]),
                                       np.asarray([self._dets[i]['scores'][self._cls]['score'] for i in idx]))
        return got


def anchor_propagate(vid_proto, track_proto, det_proto, class_idx):
    """Contract of :353-383: a tubelet's anchor is its one box with ``anchor == 0``; the detection of that frame that
    overlaps the anchor box most (first one on ties) lends its ``class_idx`` score to EVERY box of the tubelet."""
    assert vid_proto['video'] == track_proto['video']
    tubelets_proto = tubelets_proto_from_tracks_proto(track_proto['tracks'], class_idx)
    logging.info("Propagating anchor scores in {} for {}...".format(vid_proto['video'],
                                                                     imagenet_vdet_classes[class_idx]))
    dets = _ClassDetsByFrame(det_proto, class_idx)
    for tubelet in tubelets_proto:
        anchors = [box for box in tubelet['boxes'] if box['anchor'] == 0]
        assert len(anchors) == 1
        det_boxes, det_scores = dets(anchors[0]['frame'])
        best = int(np.argmax(iou([anchors[0]['bbox']], det_boxes)[0]))
        for box in tubelet['boxes']:
            box['det_score'] = det_scores[best]
    return {'video': vid_proto['video'], 'method': "anchor_propagate", 'tubelets': tubelets_proto}


def score_proto_temporal_maxpool(score_proto, window_size):
    """:386-414 -- centred sliding max of det_score per tubelet (out-of-range = -1e5), written back
    into the SAME box dicts; the returned proto is a shallow copy whose method gets the suffix
    '_temporal_maxpool_{w}'.  Raises on an even window and on ground-truth tubelets (after having
    processed the tubelets before it, like the reference)."""
    if window_size == 1:
        return score_proto
    if window_size % 2 != 1:
        raise ValueError('Window size must be odd!')
    new_score_proto = copy.copy(score_proto)
    new_score_proto['method'] += '_temporal_maxpool_{}'.format(window_size)
    tubelets = new_score_proto['tubelets']
    n_ok = len(tubelets)
    for t, tubelet in enumerate(tubelets):
        if tubelet['gt'] == 1:
            n_ok = t
            break
    series = [[box['det_score'] for box in tubelet['boxes']] for tubelet in tubelets[:n_ok]]
    pooled = hot.series_maxpool(series, window_size, -1e+5)
    for tubelet, new in zip(tubelets[:n_ok], pooled):
        for box, v in zip(tubelet['boxes'], new):
            box['det_score'] = float(v)
    if n_ok < len(tubelets):
        raise ValueError('Dangerous: Score file contains gt tracks!')
    return new_score_proto


def extrap1d(interpolator):
    """:416-428 -- point-wise linear extrapolation wrapper around a scipy interp1d-like object
    (kept for API compatibility; score_proto_interpolation does not need it here)."""
    xs, ys = interpolator.x, interpolator.y

    def pointwise(x):
        if x < xs[0]:
            return ys[0] + (x - xs[0]) * (ys[1] - ys[0]) / (xs[1] - xs[0])
        elif x > xs[-1]:
            return ys[-1] + (x - xs[-1]) * (ys[-1] - ys[-2]) / (xs[-1] - xs[-2])
        return interpolator(x)
    return pointwise


def score_proto_interpolation(score_proto, vid_proto):
    """:430-490 -- per tubelet with >= 2 boxes: linear interpolation of x1,y1,x2,y2,det_score,
    anchor to every integer frame of [min, max] (min 2 -> 1 and max F-1 -> F by one-step
    extrapolation, :472-475).  Output boxes carry only frame, det_score, anchor, bbox."""
    new_score_proto = {'video': score_proto['video'], 'method': score_proto['method'] + '_interpolation'}
    max_frames = len(vid_proto['frames'])
    out = []
    jobs = []
    for tubelet in score_proto['tubelets']:
        if tubelet['gt'] == 1:
            raise ValueError('Dangerous: Score file contains gt tracks!')
        if len(tubelet['boxes']) < 2:
            out.append(copy.copy(tubelet))
            continue
        truth_idx = [box['frame'] for box in tubelet['boxes']]
        fields = np.asarray([[box['bbox'][0], box['bbox'][1], box['bbox'][2], box['bbox'][3],
                              box['det_score'], box['anchor']] for box in tubelet['boxes']], dtype=np.float64)
        order = np.argsort(np.asarray(truth_idx), kind='stable')      # interp1d sorts its knots
        min_idx, max_idx = min(truth_idx), max(truth_idx)
        if min_idx == 2:
            min_idx = 1
        if max_idx == max_frames - 1:
            max_idx = max_frames
        new_tubelet = {}
        for key in ['gt', 'class', 'class_index']:
            new_tubelet[key] = tubelet[key]
        new_tubelet['boxes'] = []
        out.append(new_tubelet)
        jobs.append((new_tubelet, np.asarray(truth_idx, dtype=np.float64)[order], fields[order],
                     list(range(min_idx, max_idx + 1))))
    dense = hot.series_interp([j[1] for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs])
    for (new_tubelet, _, _, frames), vals in zip(jobs, dense):
        for dense_idx, v in zip(frames, vals):
            new_tubelet['boxes'].append({'frame': dense_idx, 'det_score': float(v[4]), 'anchor': float(v[5]),
                                         'bbox': [float(v[0]), float(v[1]), float(v[2]), float(v[3])]})
    new_score_proto['tubelets'] = out
    return new_score_proto


def raw_dets_spatial_max_pooling(vid_proto, track_proto, frame_to_det, class_idx, overlap_thres=0.7):
    """:493-535 -- detections as per-frame arrays {frame_id: (boxes [B,4], zs [B,C])}; class
    column zs[:, class_idx - 1] (:514).  Frames missing from frame_to_det or without boxes are
    skipped (:511-513)."""
    assert vid_proto['video'] == track_proto['video']
    score_proto = {'video': vid_proto['video'],
                   'method': "spatial_max_pooling_IOU_{}".format(overlap_thres)}
    tubelets_proto = tubelets_proto_from_tracks_proto(track_proto['tracks'], class_idx)
    logging.info("Sampling dets in {} for {}...".format(vid_proto['video'], imagenet_vdet_classes[class_idx]))
    frame_dets = {}
    for frame_id, (det_boxes, det_scores) in frame_to_det.items():
        if np.asarray(det_boxes).size == 0:
            continue
        frame_dets[frame_id] = (np.asarray(det_boxes), np.asarray(det_scores)[:, class_idx - 1].ravel())
    _spatial_max_pooling(vid_proto, tubelets_proto, frame_dets, overlap_thres)
    score_proto['tubelets'] = tubelets_proto
    do_score_completion(score_proto)
    return score_proto
