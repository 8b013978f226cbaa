"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

All generators use ``np.random.RandomState(seed)`` (the legacy stream is frozen across numpy
versions) so inputs can be regenerated from seeds on the GPU box; only seeds + reference outputs
are committed under tests/golden/.
"""
import numpy as np

W, H = 1280, 720


def boxes_1(rng, n, frac=False, degenerate=0):
    """n boxes in a 1280x720 image, SURVEY appendix A recipe (x1,y1,w,h drawn in that order)."""
    x1 = rng.uniform(0, W - 50, n)
    y1 = rng.uniform(0, H - 50, n)
    w = rng.uniform(10, 300, n)
    h = rng.uniform(10, 300, n)
    b = np.stack([x1, y1, np.minimum(x1 + w, W - 1), np.minimum(y1 + h, H - 1)], 1)
    if not frac:
        b = np.round(b)
    b = b.astype(np.float32)
    if degenerate:
        # clustered near-duplicates: high IoU neighbourhoods (dense suppression chains)
        k = min(degenerate, n)
        src = rng.randint(0, n, k)
        dst = rng.randint(0, n, k)
        jit = rng.randint(-3, 4, (k, 4)).astype(np.float32)
        b[dst] = b[src] + jit
        b[:, 2] = np.maximum(b[:, 2], b[:, 0])
        b[:, 3] = np.maximum(b[:, 3], b[:, 1])
    return b


def tie_free_scores(rng, n, kind="perm"):
    """n distinct float32 scores."""
    if kind == "perm":
        return ((rng.permutation(n) + 0.5) / max(n, 1)).astype(np.float32)
    s = rng.randn(n).astype(np.float32)
    # de-duplicate by rank jitter, then verify
    order = np.argsort(s, kind="stable")
    s2 = s.copy()
    for _ in range(8):
        if len(np.unique(s2)) == n:
            break
        s2 = s + (np.argsort(order).astype(np.float32) * np.float32(2.0 ** -18))
    assert len(np.unique(s2)) == n
    return s2


def dets5(seed, n, frac=False, degenerate=0, kind="perm"):
    rng = np.random.RandomState(seed)
    b = boxes_1(rng, n, frac, degenerate)
    s = tie_free_scores(rng, n, kind)
    return np.hstack([b, s[:, None]]).astype(np.float32)


def dets6(seed, n, n_frames, frac=False, kind="perm"):
    """(frame,x1,y1,x2,y2,score) rows, frames 1..n_frames assigned at random, scores distinct
    over the whole array."""
    rng = np.random.RandomState(seed)
    b = boxes_1(rng, n, frac)
    s = tie_free_scores(rng, n, kind)
    f = rng.randint(1, n_frames + 1, n).astype(np.float32)
    return np.hstack([f[:, None], b, s[:, None]]).astype(np.float32)


def video(seed, F, B, C, frac=False, kind="perm"):
    """boxes [F,B,4] f32, scores [F,B,C] f32 tie-free per (frame, class)."""
    rng = np.random.RandomState(seed)
    boxes = np.stack([boxes_1(rng, B, frac) for _ in range(F)], 0)
    if kind == "perm":
        # one argsort of uniform noise per (f, c) column = a random permutation per column
        r = rng.rand(F, B, C)
        ranks = np.argsort(np.argsort(r, axis=1), axis=1)
        scores = ((ranks + 0.5) / B).astype(np.float32)
    else:
        scores = rng.randn(F, B, C).astype(np.float32)
    return boxes, scores


# ---------------------------------------------------------------------------------------------
# protocol-dict inputs (small), built with plain python so that the golden generator (reference
# side) and the tests (product side) start from byte-identical dicts
# ---------------------------------------------------------------------------------------------
import hashlib


def _hash(video, frame, bbox):
    return hashlib.md5('{}_{}_{}_{}_{}_{}'.format(
        video, frame, bbox[0], bbox[1], bbox[2], bbox[3]).encode()).hexdigest()


def make_vid_proto(name, F):
    return {'video': name, 'root_path': '/synthetic/' + name,
            'frames': [{'frame': i + 1, 'path': '%06d.JPEG' % i} for i in range(F)]}


def make_box_proto(seed, name, F, B):
    rng = np.random.RandomState(seed)
    boxes = []
    for f in range(1, F + 1):
        for b in boxes_1(rng, B):
            bb = [int(v) for v in b]
            boxes.append({'frame': f, 'bbox': bb, 'hash': _hash(name, f, bb)})
    return {'video': name, 'boxes': boxes}


def make_det_proto(seed, name, F, B, class_names, kind="perm"):
    """det_proto with one score dict per class (class_index = position, utils/protocol.py:307-320).
    Scores are distinct over the whole video per class."""
    rng = np.random.RandomState(seed)
    C = len(class_names)
    dets = []
    # temporally coherent proposals: frame f = frame 1 drifting by (4,2) px/frame + jitter, so
    # that stub_track_rows' drifting boxes overlap detections in neighbouring frames
    base = boxes_1(rng, B)
    all_boxes = []
    for f in range(F):
        jit = rng.randint(-2, 3, (B, 4)).astype(np.float32)
        all_boxes.append(base + np.float32(f) * np.array([4, 2, 4, 2], np.float32) + jit)
    cols = [tie_free_scores(rng, F * B, kind) for _ in range(C)]
    n = 0
    for f in range(1, F + 1):
        for b in all_boxes[f - 1]:
            bb = [int(v) for v in b]
            dets.append({'frame': f, 'bbox': bb, 'hash': _hash(name, f, bb),
                         'scores': [{'class': class_names[c], 'class_index': c,
                                     'score': float(cols[c][n])} for c in range(C)]})
            n += 1
    return {'video': name, 'detections': dets}


def stub_track_rows(n_frames, anchor_frame, bbox, span=3, drift=(4, 2), nan_at=None):
    """Deterministic stand-in for the external MATLAB tracker (vdet/track.py:52-106): the anchor
    box drifting linearly over [anchor-span, anchor+span], score 1/(1+|d|).  Returns
    (rows [L,5] float64, start_frame)."""
    start = max(1, anchor_frame - span)
    end = min(n_frames, anchor_frame + span)
    rows = []
    for f in range(start, end + 1):
        d = f - anchor_frame
        if nan_at is not None and d == nan_at:
            rows.append([float('nan')] * 5)
            continue
        rows.append([bbox[0] + drift[0] * d, bbox[1] + drift[1] * d,
                     bbox[2] + drift[0] * d, bbox[3] + drift[1] * d, 1.0 / (1 + abs(d))])
    return np.asarray(rows, dtype=np.float64), start


# ---------------------------------------------------------------------------------------------
# the protocol-level golden case (shared by tests/golden/make_golden.py and the tests)
# ---------------------------------------------------------------------------------------------
CLS5 = ['__background__', 'airplane', 'antelope', 'bear', 'bicycle']


def proto_case():
    name = 'synth_vid_a'
    F, B = 6, 40
    vid = make_vid_proto(name, F)
    det = make_det_proto(501, name, F, B, CLS5)
    det_info = np.asarray([[d['frame']] + d['bbox'] + [s['score'] for s in d['scores'][1:]]
                           for d in det['detections']], dtype=np.float64)
    frame_to_det = {}
    for f in range(1, F + 1):
        ds = [d for d in det['detections'] if d['frame'] == f]
        if f == 4:
            continue                                            # a frame without a det file
        frame_to_det[f] = (np.asarray([d['bbox'] for d in ds], dtype=np.float64) + 0.25,
                           np.asarray([[s['score'] for s in d['scores'][1:]] for d in ds], dtype=np.float32))
    frame_to_det[5] = (np.zeros((0, 4)), np.zeros((0, 4), np.float32))    # an empty frame
    annot = {'video': name, 'annotations': [
        {'id': 0, 'track': [{'frame': f, 'bbox': [10 + f, 20, 110 + f, 140], 'class': 'airplane',
                             'class_index': 1, 'name': 'n', 'occluded': 0, 'generated': 0}
                            for f in range(1, 5)]},
        {'id': 1, 'track': [{'frame': f, 'bbox': [300, 200 + f, 420, 330 + f], 'class': 'bear',
                             'class_index': 3, 'name': 'n', 'occluded': 0, 'generated': 0}
                            for f in range(2, 7)]}]}
    return dict(name=name, F=F, B=B, vid=vid, det=det, det_info=det_info, frame_to_det=frame_to_det,
                annot=annot)


def det_fun_case(Cv=4):
    """Deterministic stand-in for the CNN of fast_rcnn_det_vid (keyed on the first proposal)."""
    def det_fun(net, im, orig_boxes):
        seed = 600 + int(orig_boxes[0][0]) + 7 * int(orig_boxes[0][1])
        r = np.random.RandomState(seed)
        scores = r.rand(len(orig_boxes), Cv + 1)
        deltas = r.uniform(-5, 5, (len(orig_boxes), 4 * (Cv + 1)))
        boxes = np.tile(orig_boxes.astype(np.float64), (1, Cv + 1)) + deltas
        return scores, boxes
    return det_fun


def make_stub_tracker(tracks_proto_from_boxes, nan_at=None, span=3):
    """track_method plug-in built on either the reference's or the build's tracks_proto_from_boxes."""
    def stub_tracker(vid_proto, anchor_frame_id, anchor_bbox, opts):
        rows, start = stub_track_rows(len(vid_proto['frames']), anchor_frame_id, list(anchor_bbox),
                                      span=span, nan_at=nan_at)
        return tracks_proto_from_boxes(rows, vid_proto['video'], anchor_frame_id, start, 1)
    return stub_tracker


class FakeTCN(object):
    """Recording pycaffe-like net for score_conv_cls: probs[1] = sigmoid(det_scores)."""

    class Blob(object):
        def __init__(self, c):
            self.shape = (1, c, 1, 1)
            self.data = np.zeros(self.shape, np.float32)

        def reshape(self, *s):
            self.shape = tuple(s)
            self.data = np.zeros(s, np.float32)

    def __init__(self):
        self.blobs = {k: FakeTCN.Blob(1) for k in ('det_scores', 'track_scores', 'anchors', 'abs_anchors',
                                                  'gt_overlaps', 'labels')}
        self.calls = []

    def forward(self):
        L = self.blobs['det_scores'].shape[3]
        self.calls.append({k: np.array(b.data).ravel().tolist() for k, b in self.blobs.items()})
        z = self.blobs['det_scores'].data.reshape(L)
        p1 = 1.0 / (1.0 + np.exp(-z))
        return {'probs': np.stack([1 - p1, p1])[None].astype(np.float32)}


def vid_with_objects(seed, F, B, C, n_obj=3):
    """Ground-truth objects drifting over the frames; proposals = jittered copies of the objects
    (scored high for the object's class) + clutter."""
    rng = np.random.RandomState(seed)
    objs = []
    for k in range(n_obj):
        x, y = rng.uniform(50, 900), rng.uniform(50, 450)
        w, h = rng.uniform(60, 250), rng.uniform(60, 220)
        objs.append(dict(cls=int(rng.randint(1, C + 1)), box=np.array([x, y, x + w, y + h]), v=rng.uniform(-4, 4, 2)))
    boxes = np.zeros((F, B, 4), np.float32)
    scores = (0.05 * rng.rand(F, B, C)).astype(np.float32)
    annot = {'video': 'syn_%d' % seed, 'annotations': []}
    for k, o in enumerate(objs):
        annot['annotations'].append({'id': str(k), 'track': []})
    for f in range(F):
        clutter_x = rng.uniform(0, 1100, B); clutter_y = rng.uniform(0, 600, B)
        boxes[f] = np.stack([clutter_x, clutter_y, clutter_x + rng.uniform(20, 200, B), clutter_y + rng.uniform(20, 150, B)], 1)
        for k, o in enumerate(objs):
            gtb = np.round(o['box'] + np.tile(o['v'], 2) * f)
            annot['annotations'][k]['track'].append({'frame': f + 1, 'bbox': [int(v) for v in gtb], 'class_index': o['cls'],
                                                     'class': 'c%d' % o['cls']})
            for j in range(6):                                   # 6 jittered proposals per object
                b = k * 6 + j
                boxes[f, b] = gtb + rng.randint(-6, 7, 4)
                scores[f, b, o['cls'] - 1] = 0.6 + 0.39 * rng.rand()
    return np.round(boxes).astype(np.float32), scores, annot


def coherent_video(seed, F, B, C, jitter=3, frac=False):
    """Proposals that persist over time (frame f = frame 0 drifting + jitter) so that IoU links continue
    from frame to frame; scores ~U(0,1) f32.  frac: fractional coordinates (int truncation matters)."""
    rng = np.random.RandomState(seed)
    base = boxes_1(rng, B)
    boxes = np.stack([base + np.float32(f) * np.array([3, 2, 3, 2], np.float32) +
                      rng.randint(-jitter, jitter + 1, (B, 4)).astype(np.float32) for f in range(F)], 0)
    if frac:
        boxes = boxes + rng.uniform(0, 0.99, boxes.shape).astype(np.float32)
    scores = rng.rand(F, B, C).astype(np.float32)
    return boxes.astype(np.float32), scores


# LINK-stage golden cases (tests/golden/make_golden.py g14 <-> tests/test_link_golden*.py)
LINK_CASES = [
    dict(name='plain', seed=1401, F=8, B=200, C=3, max_tracks=4, thres=0.0, max_frames=0, nms_thres=0.3, link=0.5, pool=0.7, window=3),
    dict(name='ties', seed=1402, F=5, B=64, C=2, max_tracks=6, thres=0.5, max_frames=0, nms_thres=0.3, link=0.5, pool=0.5, window=3, frame_ranks=True),
    dict(name='max_frames', seed=1403, F=12, B=300, C=2, max_tracks=5, thres=0.0, max_frames=5, nms_thres=0.3, link=0.5, pool=0.7, window=5),
    dict(name='thres_stop', seed=1404, F=9, B=250, C=2, max_tracks=12, thres=0.997, max_frames=0, nms_thres=0.3, link=0.5, pool=0.7, window=3),
    dict(name='frac', seed=1405, F=7, B=180, C=2, max_tracks=4, thres=0.0, max_frames=0, nms_thres=0.4, link=0.45, pool=0.6, window=3, frac=True),
    dict(name='multi_class', seed=1406, F=10, B=500, C=4, max_tracks=3, thres=0.0, max_frames=0, nms_thres=0.3, link=0.5, pool=0.7, window=7),
    dict(name='incoherent', seed=1407, F=6, B=400, C=2, max_tracks=5, thres=0.0, max_frames=0, nms_thres=0.3, link=0.5, pool=0.7, window=3, random=True),
    dict(name='loose_link', seed=1408, F=8, B=150, C=2, max_tracks=12, thres=0.0, max_frames=3, nms_thres=0.5, link=0.2, pool=0.3, window=3),
]


def link_case_video(case):
    if case.get('random'):
        boxes, scores = video(case['seed'], case['F'], case['B'], case['C'], kind='randn')
        scores = (1.0 / (1.0 + np.exp(-scores))).astype(np.float32)
    else:
        boxes, scores = coherent_video(case['seed'], case['F'], case['B'], case['C'], frac=bool(case.get('frac')))
    if case.get('frame_ranks'):
        # every frame holds the SAME set of score values (ranks / B): the anchor order ties across frames at every
        # level (vdet/track.py:200's stable sort decides: lowest flat index first) while no two detections of one
        # frame tie -- ties inside a frame would hit the reference's unstable argsort in vid_nms (utils/nms.pyx:80),
        # which no implementation can reproduce (hard part 2)
        ranks = np.argsort(np.argsort(scores, axis=1), axis=1)
        scores = ((ranks + 0.5) / scores.shape[1]).astype(np.float32)
    return boxes, scores


# Exotic inputs (NaN / inf coordinates and scores, zero-area boxes, odd thresholds): what the REFERENCE does with them is
# recorded by tests/golden/make_golden.py --exotic-only into tests/golden/exotic_golden.npz.  Scores stay tie-free
# (one NaN score at most: several would tie in the reference's unstable argsort).
EXOTIC_CASES = [
    dict(name='nan_x1', fn='nms', seed=1501, n=300, thresh=0.3, nan_cols=[0], nan_rows=7),
    dict(name='nan_y2_and_score', fn='nms', seed=1502, n=257, thresh=0.5, nan_cols=[3], nan_rows=5, nan_score=True),
    dict(name='inf_coords', fn='nms', seed=1503, n=200, thresh=0.3, inf_rows=3),
    dict(name='randn_scores_one_nan', fn='nms', seed=1504, n=400, thresh=0.3, nan_score=True, kind='randn'),
    dict(name='zero_area', fn='nms', seed=1505, n=120, thresh=0.3, zero_area=6),
    dict(name='thresh_nan', fn='nms', seed=1506, n=150, thresh=float('nan')),
    dict(name='thresh_negative', fn='nms', seed=1507, n=150, thresh=-1.0),
    dict(name='identical_boxes', fn='nms', seed=1508, n=64, thresh=0.3, identical=True),
    dict(name='vid_nan_frames', fn='vid_nms', seed=1509, n=500, n_frames=4, thresh=0.3, nan_frames=5),
    dict(name='vid_nan_box', fn='vid_nms', seed=1510, n=300, n_frames=2, thresh=0.3, nan_cols=[1], nan_rows=4),
    dict(name='tdn_nan_track', fn='track_det_nms', seed=1511, m=300, t=3, n_frames=2, thresh=0.3, nan_track=True),
    dict(name='tdn_nan_det', fn='track_det_nms', seed=1512, m=300, t=2, n_frames=2, thresh=0.3, nan_cols=[2], nan_rows=6),
]


def exotic_inputs(case):
    """(dets [, tracks]) float32 arrays of an EXOTIC_CASES entry."""
    rng = np.random.RandomState(case['seed'])
    if case['fn'] == 'nms':
        d = dets5(case['seed'], case['n'], kind=case.get('kind', 'perm'))
        off = 0
    else:
        n = case.get('n', case.get('m'))
        d = dets6(case['seed'], n, case['n_frames'])
        off = 1
    pick = rng.permutation(len(d))
    for c in case.get('nan_cols', []):
        d[pick[:case['nan_rows']], c] = np.nan            # column index as given (vid / track rows: 0 = frame, 1..4 = box)
    if case.get('nan_score'):
        d[pick[-1], off + 4] = np.nan
    if case.get('inf_rows'):
        k = case['inf_rows']
        d[pick[:k], off + 2] = np.inf
        d[pick[k:k + 2], off + 0] = -np.inf
    if case.get('zero_area'):
        k = case['zero_area']
        d[pick[:k], off + 2] = d[pick[:k], off + 0] - 1.0          # width 0 (+1 convention)
    if case.get('identical'):
        d[:, off:off + 4] = d[0, off:off + 4]
    if case.get('nan_frames'):
        d[pick[:case['nan_frames']], 0] = np.nan
    if case['fn'] != 'track_det_nms':
        return (d,)
    tb = boxes_1(rng, case['t'])
    tf = rng.randint(1, case['n_frames'] + 1, case['t']).astype(np.float32)
    tr = np.hstack([tf[:, None], tb]).astype(np.float32).reshape(-1, 5)
    if case.get('nan_track'):
        tr[0, 2] = np.nan
    return (tr, d)


# ---------------------------------------------------------------------------------------------
# per-frame .mat detection files (utils/protocol.py:528-555: load_frame_to_det / load_det_info)
# ---------------------------------------------------------------------------------------------
MAT_CASE = dict(seed=1601, name='synth_vid_mat', F=7, C=5)


def write_mat_case(det_dir):
    """Writes the case's per-frame ``.mat`` files (``boxes [B,4]``, ``zs [B,C]``, both float64 as MATLAB saves them)
    into det_dir and returns the vid_proto.  Frame 1..F: B = 6, 0 (empty arrays), 3, -- (no file), 9, 4, 1; frame 5's
    file is named after the full frame path (``<path>.mat``, the loaders' second guess), the others after its stem."""
    import os
    import scipy.io as sio
    c = MAT_CASE
    rng = np.random.RandomState(c['seed'])
    vid = make_vid_proto(c['name'], c['F'])
    counts = [6, 0, 3, None, 9, 4, 1]
    for frame, B in zip(vid['frames'], counts):
        if B is None:
            continue
        boxes = boxes_1(rng, B, frac=True).astype(np.float64).reshape(B, 4)
        zs = rng.randn(B, c['C'])
        stem = os.path.splitext(frame['path'])[0]
        name = (frame['path'] if frame['frame'] == 5 else stem) + '.mat'
        path = os.path.join(det_dir, name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        sio.savemat(path, {'boxes': boxes, 'zs': zs})
    return vid
