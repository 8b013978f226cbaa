"""Protocol dicts: the data model T-CNN scripts program against (reference utils/protocol.py).

Plain dicts/lists serialised as JSON (optionally .gz).  Schemas (reference docstring :7-192):

  .vid    {'video', 'root_path', 'frames': [{'frame' (1-based), 'path'}]}
  .box    {'video', 'boxes': [{'frame', 'bbox': [x1,y1,x2,y2], 'hash'}]}
  .det    {'video', 'detections': [{'frame','bbox','hash','scores': [{'class','class_index','score'}]}]}
  .track  {'video', 'method', 'tracks': [[{'frame','bbox','hash','score','anchor'}, ...], ...]}
  .score  {'video', 'method', 'tubelets': [{'gt','class','class_index',
                                            'boxes': [{'frame','bbox','hash','anchor','track_score',
                                                       'det_score', (...)}]}]}
  .annot  {'video', 'annotations': [{'id', 'track': [{'frame','bbox','class','class_index',...}]}]}

Every function keeps the reference's name, argument order, defaults, dict keys and its in-place
mutation behaviour (callers rely on it); line numbers cite /root/reference/utils/protocol.py.
This is host logic: no numeric kernel lives here except ``tubelets_overlap`` (IoU on the GPU via
``common.iou``).  c2-scale data does not travel as dicts -- see ``vdetlib_amd.ops``.
"""
import copy
import gzip
import hashlib
import json
import os

import numpy as np

from .common import isimg, sort_nicely, iou, stem
from .log import logger as logging
from ..vdet.dataset import imagenet_vdet_classes


# ------------------------------------------------------------------------------------------------
# general
# ------------------------------------------------------------------------------------------------

def proto_load(file_path):
    """:209-220 -- a sibling ``file_path + '.gz'`` wins silently when it exists."""
    if os.path.isfile(file_path + '.gz'):
        file_path += '.gz'
    if os.path.splitext(file_path)[1] == '.gz':
        with gzip.GzipFile(file_path) as f:
            return json.loads(f.read().decode('utf-8'))
    with open(file_path, 'r') as f:
        return json.load(f)


def _json_text(obj):
    return json.dumps(obj, indent=2)


def proto_dump(obj, file_path):
    """Contract of :223-236: the protocol as JSON text with two-space indentation; a name ending in ``.gz`` is written
    gzip-compressed at level 1 -- unless gzip refuses the buffer (OverflowError: more than it can frame), in which case
    whatever was started is removed and the text goes, uncompressed, to the same name without ``.gz``."""
    text = _json_text(obj)
    stem_path, ext = os.path.splitext(file_path)
    target = file_path
    if ext == '.gz':
        try:
            with gzip.GzipFile(file_path, mode='w', compresslevel=1) as gz:
                gz.write(text.encode('utf-8'))
            return
        except OverflowError:
            print("{}: too large for gzip, writing {} uncompressed instead.".format(file_path, stem_path))
            if os.path.isfile(file_path):
                os.remove(file_path)
            target = stem_path
    with open(target, 'w') as out:
        out.write(text)


# ------------------------------------------------------------------------------------------------
# video protocol
# ------------------------------------------------------------------------------------------------

def vid_proto_from_dir(root_dir, vid_name=None):
    """:243-257.  (The reference forgets to import ``stem`` and raises NameError when vid_name is
    None; here the name is inferred from the directory as intended.)"""
    frame_list = [i for i in os.listdir(root_dir) if isimg(i)]
    sort_nicely(frame_list)
    vid = {'root_path': root_dir,
           'frames': [{'frame': index + 1, 'path': path} for index, path in enumerate(frame_list)]}
    if not vid_name:
        vid_name = stem(root_dir)
    vid['video'] = vid_name
    return vid


def frame_path_at(vid_proto, frame_id):
    """:260-262 (IndexError when the frame does not exist, like the reference)."""
    frame = [frame for frame in vid_proto['frames'] if frame['frame'] == frame_id][0]
    return str(os.path.join(vid_proto['root_path'], frame['path']))


def frame_path_before(vid_proto, frame_id):
    return [str(os.path.join(vid_proto['root_path'], frame['path']))
            for frame in vid_proto['frames'] if frame['frame'] <= frame_id]


def frame_path_after(vid_proto, frame_id):
    return [str(os.path.join(vid_proto['root_path'], frame['path']))
            for frame in vid_proto['frames'] if frame['frame'] >= frame_id]


def sample_vid_proto(vid_proto, stride=10):
    """:277-284 -- every stride-th frame entry (same dict objects)."""
    logging.info("Sampling video by 1 / {}.".format(stride))
    return {'video': vid_proto['video'], 'root_path': vid_proto['root_path'],
            'frames': [vid_proto['frames'][i] for i in range(0, len(vid_proto['frames']), stride)]}


def path_to_index(vid_proto, path):
    for frame in vid_proto['frames']:
        if frame['path'].startswith(path):
            return frame['frame']
    return None


# ------------------------------------------------------------------------------------------------
# detection protocol
# ------------------------------------------------------------------------------------------------

def empty_det_from_box(box_proto):
    """:297-304 -- REUSES the box dicts (adds 'scores': [] to each, in place)."""
    detections = box_proto['boxes']
    for det in detections:
        det['scores'] = []
    return {'video': box_proto['video'], 'detections': detections}


def score_proto(class_names, scores):
    """:307-320 -- class_index is the POSITION in class_names."""
    if type(scores) is not list:
        scores = scores.tolist()
    return [{'class': cls_name, 'class_index': idx, 'score': score}
            for idx, (cls_name, score) in enumerate(zip(class_names, scores))]


def det_score(detection, class_index):
    """:323-327 -- looked up BY KEY; -inf when the class is missing."""
    for score in detection['scores']:
        if score['class_index'] == class_index:
            return score['score']
    return float('-inf')


def top_detections(det_proto, top_num, class_index):
    """:330-339"""
    if len(det_proto['detections']) < top_num:
        return copy.copy(det_proto)
    sorted_det = sorted(copy.copy(det_proto['detections']),
                        key=lambda x: det_score(x, class_index), reverse=True)
    return {'video': det_proto['video'], 'detections': sorted_det[:top_num]}


def frame_top_detections(det_proto, top_num, class_index):
    """:341-351 (frame order follows python's set iteration, as in the reference)."""
    new_det = {'video': det_proto['video'], 'detections': []}
    for frame_id in list(set(det['frame'] for det in det_proto['detections'])):
        cur_dets = sorted([det for det in det_proto['detections'] if det['frame'] == frame_id],
                          key=lambda x: det_score(x, class_index), reverse=True)
        new_det['detections'].extend(cur_dets[:top_num])
    return new_det


# ------------------------------------------------------------------------------------------------
# proposal protocol
# ------------------------------------------------------------------------------------------------

def bbox_hash(video_name, frame_id, bbox):
    """:372-375 -- md5 of "video_frame_x1_y1_x2_y2" (the un-truncated values)."""
    return hashlib.md5('{}_{}_{}_{}_{}_{}'.format(
        video_name, frame_id, bbox[0], bbox[1], bbox[2], bbox[3]).encode('utf-8')).hexdigest()


def boxes_proto_from_boxes(frame_idx_list, boxes_list, video_name):
    """:358-369 -- returns the LIST of box dicts."""
    boxes_proto = []
    for frame_idx, boxes in zip(frame_idx_list, boxes_list):
        for bbox in boxes:
            boxes_proto.append({'frame': int(frame_idx), 'bbox': [int(v) for v in bbox],
                                'hash': bbox_hash(video_name, frame_idx, bbox)})
    return boxes_proto


def boxes_at_frame(box_proto, frame_id):
    """:378-383 -- shallow copies of the boxes of one frame."""
    return [copy.copy(box) for box in box_proto['boxes'] if box['frame'] == frame_id]


# ------------------------------------------------------------------------------------------------
# tracking protocol
# ------------------------------------------------------------------------------------------------

def tracks_proto_from_boxes(boxes, video_name, anchor, start_frame=1, step=1):
    """:389-414 -- rows (x1,y1,x2,y2,score); a row with any NaN ends the current tracklet;
    bbox ints by truncation, hash over the un-truncated row, anchor = int(frame - anchor)."""
    tracks_proto = []
    track = None
    for box_idx, bbox in enumerate(boxes):
        frame_idx = start_frame + box_idx * step
        if np.any(np.isnan(bbox)):
            if track is not None:
                tracks_proto.append(track)
                track = None
            continue
        if track is None:
            track = []
        track.append({'frame': frame_idx,
                      'bbox': [int(cor) for cor in bbox[0:4]],
                      'hash': bbox_hash(video_name, frame_idx, bbox),
                      'score': float(bbox[4]),
                      'anchor': int(frame_idx - anchor)})
    if track is not None:
        tracks_proto.append(track)
    return tracks_proto


def track_box_at_frame(tracklet, frame_id):
    for box in tracklet:
        if box['frame'] == frame_id:
            return box['bbox']
    return None


def track_proto_from_annot_proto(annot_proto):
    """:423-443 -- ground-truth tracks (score 1, anchor 0, method 'gt')."""
    vid_name = annot_proto['video']
    tracks = []
    for annot_track in annot_proto['annotations']:
        tracks.append([{"frame": b['frame'], "bbox": b['bbox'], "score": 1, "anchor": 0,
                        "hash": bbox_hash(vid_name, b['frame'], b['bbox'])}
                       for b in annot_track['track']])
    return {'video': vid_name, 'method': 'gt', 'tracks': tracks}


# ------------------------------------------------------------------------------------------------
# scoring protocol
# ------------------------------------------------------------------------------------------------

def tubelets_proto_from_tracks_proto(tracks_proto, class_index):
    """:448-464 -- 'score' becomes 'track_score', det_score starts at the -1e5 sentinel."""
    tubelets = []
    for track in tracks_proto:
        boxes = []
        for box in track:
            tb = copy.copy(box)
            tb['track_score'] = tb['score']
            tb['det_score'] = -1e5
            del tb['score']
            boxes.append(tb)
        tubelets.append({'gt': 0, 'class_index': class_index,
                         'class': imagenet_vdet_classes[class_index], 'boxes': boxes})
    return tubelets


def _annotated_boxes_by_frame(annot_proto, class_index):
    """frame -> ground-truth bboxes of ``class_index``.  A track contributes its boxes up to (not including) the first
    one of another class -- the reference leaves a track at the first class mismatch (:474-478)."""
    by_frame = {}
    for annot_track in annot_proto['annotations']:
        for annot_box in annot_track['track']:
            if annot_box['class_index'] != class_index:
                break
            by_frame.setdefault(annot_box['frame'], []).append(annot_box['bbox'])
    return by_frame


def tubelets_overlap(tubelets_proto, annot_proto, class_idx):
    """Contract of :467-489, in place: every tubelet box gets ``gt_overlap`` = its best IoU with a same-frame
    ground-truth box of the tubelet's class (the integer 0 when nothing overlaps), and a tubelet whose boxes all coincide
    with ground truth (mean overlap 1 up to float eps) is flagged ``gt = 1``.  ``class_idx`` is unused, as in the
    reference (each tubelet carries its own class)."""
    gt_cache = {}
    for tubelet in tubelets_proto:
        cls = tubelet['class_index']
        if cls not in gt_cache:
            gt_cache[cls] = _annotated_boxes_by_frame(annot_proto, cls)
        overlaps = []
        for tubelet_box in tubelet['boxes']:
            best = 0
            for gt_bbox in gt_cache[cls].get(tubelet_box['frame'], ()):
                value = float(iou([gt_bbox], [tubelet_box['bbox']]).ravel()[0])
                if value > best:
                    best = value
            tubelet_box['gt_overlap'] = best
            overlaps.append(best)
        if abs(np.asarray(overlaps).mean() - 1) < np.finfo(float).eps:
            tubelet['gt'] = 1
    return tubelets_proto


def tubelet_box_at_frame(tubelet, frame_id):
    for box in tubelet['boxes']:
        if box['frame'] == frame_id:
            return box['bbox']
    return None


def tubelet_box_proto_at_frame(tubelet, frame_id):
    for box in tubelet['boxes']:
        if box['frame'] == frame_id:
            return box
    return None


def _assert_same(first, second, keys):
    for key in keys:
        assert first[key] == second[key]


def merge_score_protos(proto_1, proto_2, scheme='combine'):
    """Contract of :504-525.  The result is a SHALLOW copy of proto_1 (its tubelet list and boxes are proto_1's own
    objects -- the reference's aliasing, which callers rely on): the method names are joined with '_' when they
    differ; 'combine' appends proto_2's tubelets to that shared list; 'max' walks both protos tubelet by tubelet and
    box by box (same gt / class / frame / anchor asserted) and, wherever proto_2's box has the higher det_score,
    overwrites the fields proto_1's box has with copies of proto_2's."""
    assert scheme in ['combine', 'max']
    assert proto_1['video'] == proto_2['video']
    merged = copy.copy(proto_1)
    methods = [proto_1['method'], proto_2['method']]
    if methods[0] != methods[1]:
        merged['method'] = '_'.join(methods)
    if scheme == 'combine':
        merged['tubelets'] += list(proto_2['tubelets'])
        return merged
    for mine, other in zip(merged['tubelets'], proto_2['tubelets']):
        _assert_same(mine, other, ('gt', 'class', 'class_index'))
        for box, rival in zip(mine['boxes'], other['boxes']):
            _assert_same(box, rival, ('frame', 'anchor'))
            if rival['det_score'] > box['det_score']:
                box.update({key: copy.copy(rival[key]) for key in list(box)})
    return merged


def _det_file_for(frame, det_dir):
    basename = os.path.splitext(frame['path'])[0]
    score_file = os.path.join(det_dir, basename + '.mat')
    if not os.path.isfile(score_file):
        score_file = os.path.join(det_dir, frame['path'] + '.mat')
    return score_file if os.path.isfile(score_file) else None


def load_frame_to_det(vid_proto, det_dir):
    """:528-539 -- {frame_id: (boxes [B,4], zs [B,C])} from per-frame .mat files."""
    import scipy.io as sio
    frame_to_det = {}
    for frame in vid_proto['frames']:
        score_file = _det_file_for(frame, det_dir)
        if score_file:
            d = sio.loadmat(score_file)
            frame_to_det[frame['frame']] = (d['boxes'], d['zs'])
    return frame_to_det


def load_det_info(vid_proto, det_dir):
    """:541-555 -- rows [frame_id, x1, y1, x2, y2, scores...] as one float array."""
    import scipy.io as sio
    det_info = []
    for frame in vid_proto['frames']:
        score_file = _det_file_for(frame, det_dir)
        if score_file:
            d = sio.loadmat(score_file)
            if d['boxes'].size == 0:
                continue
            for boxes, scores in zip(d['boxes'], d['zs']):
                det_info.append([frame['frame']] + boxes.tolist() + scores.tolist())
    return np.asarray(det_info)
