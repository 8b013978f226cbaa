"""BASELINE configs[0] end to end through the DICT API, the way T-CNN drives it: 30 frames x 300 proposals x 30 classes,

    fast_rcnn_det_vid (threshold 0.05 + top-100, vdet/video_det.py:64-106)
    -> apply_image_nms per (frame, class)            (vdet/image_det.py:117-123,  900 calls)
    -> apply_vid_nms per class                       (vdet/video_det.py:51-61,     30 calls, 9 000 detections each)
    -> greedily_track_from_raw_dets per class        (vdet/track.py:189-252, 10 tracks, one track_det_nms per tracked box)
    -> raw_dets_spatial_max_pooling per class        (vdet/tubelet_cls.py:493-535 incl. do_score_completion :284-303)
    -> score_proto_temporal_maxpool(3) per class     (vdet/tubelet_cls.py:386-414)

The SAME driver runs on the reference's modules (tests/golden/make_golden.py --c1, build container only: it records the
outputs in tests/golden/c1_flow_golden.json.gz and the reference's seconds per function in oracle/reference_c1.json) and on
the build's (`vdetlib.*` names: tests/test_c1_flow_gpu.py, bench.py's c1_dict_api leg).  Inputs are seeded (tests/synth.py).
Test infrastructure; nothing here is imported by the product."""
import contextlib
import hashlib
import io
import time

import numpy as np

import synth

F, B, C = 30, 300, 30
NAME = 'c1_vid'
CLASS_NAMES = ['__background__'] + ['class_%02d' % i for i in range(1, C + 1)]
MAX_TRACKS, TRACK_THRES, NMS_THRES, POOL_THRES, WINDOW, SPAN = 10, 0.5, 0.3, 0.7, 3, 5


def inputs():
    """everything the flow consumes, built with plain python / numpy from seeds"""
    vid = synth.make_vid_proto(NAME, F)
    box = synth.make_box_proto(1000, NAME, F, B)
    det = synth.make_det_proto(1001, NAME, F, B, CLASS_NAMES)
    det_info = np.asarray([[d['frame']] + d['bbox'] + [s['score'] for s in d['scores'][1:]] for d in det['detections']],
                          dtype=np.float64)                                       # [F*B, 5 + C]: vdet/track.py:189-201
    frame_to_det = {}
    for f in range(1, F + 1):
        rows = det_info[det_info[:, 0] == f]
        frame_to_det[f] = (rows[:, 1:5].copy(), rows[:, 5:].astype(np.float32))   # boxes [B,4], zs [B,C] (protocol.py:528-539)
    return dict(vid=vid, box=box, det=det, det_info=det_info, frame_to_det=frame_to_det)


def run(mods, inp=None, classes=None):
    """mods: dict(V=video_det, I=image_det, K=track, T=tubelet_cls, P=protocol, Cm=common) of either implementation.
    Returns (seconds per function, outputs as plain python data)."""
    V, I, K, T, P, Cm = (mods[k] for k in ('V', 'I', 'K', 'T', 'P', 'Cm'))
    inp = inp or inputs()
    vid, box, det, det_info, f2d = (inp[k] for k in ('vid', 'box', 'det', 'det_info', 'frame_to_det'))
    classes = list(classes or range(1, C + 1))
    sec, out = {}, {}
    sink = io.StringIO()
    V.imread = lambda p: None
    det_fun = synth.det_fun_case(C)
    with contextlib.redirect_stdout(sink):
        t = time.perf_counter()
        all_boxes = V.fast_rcnn_det_vid(None, vid, box, det_fun, class_names=CLASS_NAMES, max_per_image=100, thresh=0.05)
        sec['fast_rcnn_det_vid'] = time.perf_counter() - t
        t = time.perf_counter()
        keeps = [[I.apply_image_nms(all_boxes[j][i][:, :4], all_boxes[j][i][:, 4], NMS_THRES) for i in range(F)] for j in classes]
        sec['apply_image_nms_x%d' % (F * len(classes))] = time.perf_counter() - t
        out['image_nms_keep'] = [[[int(k) for k in kk] for kk in row] for row in keeps]
        out['fast_rcnn_rows'] = [[int(len(all_boxes[j][i])) for i in range(F)] for j in classes]
        t = time.perf_counter()
        vn = [V.apply_vid_nms(det, ci) for ci in classes]
        sec['apply_vid_nms_x%d' % len(classes)] = time.perf_counter() - t
        # (thousands of kept detections per class: their count + one digest of their hashes, in order)
        out['vid_nms_kept'] = [[len(v['detections']), hashlib.md5(','.join(d['hash'] for d in v['detections']).encode()).hexdigest()] for v in vn]
        trk = synth.make_stub_tracker(P.tracks_proto_from_boxes, span=SPAN)
        t = time.perf_counter()
        tracks = []
        for ci in classes:
            opts = Cm.options({'max_tracks': MAX_TRACKS, 'thres': TRACK_THRES, 'nms_thres': NMS_THRES})
            tracks.append(K.greedily_track_from_raw_dets(vid, det_info, trk, ci, opts))
        sec['greedily_track_from_raw_dets_x%d' % len(classes)] = time.perf_counter() - t
        out['tracks'] = [[[[int(b['frame'])] + [int(v) for v in b['bbox']] for b in tr] for tr in tp['tracks']] for tp in tracks]
        t = time.perf_counter()
        sps = [T.raw_dets_spatial_max_pooling(vid, tp, f2d, ci, POOL_THRES) for tp, ci in zip(tracks, classes)]
        sec['raw_dets_spatial_max_pooling_x%d' % len(classes)] = time.perf_counter() - t
        out['spatial'] = [[[[int(b['frame']), float(b['det_score'])] + [float(v) for v in b['bbox']] for b in tub['boxes']]
                           for tub in sp['tubelets']] for sp in sps]
        t = time.perf_counter()
        tps = [T.score_proto_temporal_maxpool(sp, WINDOW) for sp in sps]
        sec['score_proto_temporal_maxpool_x%d' % len(classes)] = time.perf_counter() - t
        out['temporal'] = [[[float(b['det_score']) for b in tub['boxes']] for tub in tp['tubelets']] for tp in tps]
        out['methods'] = [tps[0]['method'], sps[0]['method'], tracks[0]['method']]
    sec['total'] = sum(sec.values())
    return sec, out


def compare(got, want, tol=1e-5):
    """integer / string results exact, float scores within tol (north_star).  Returns the list of mismatching keys."""
    bad = []

    def same(a, b):
        if isinstance(a, (list, tuple)):
            return isinstance(b, (list, tuple)) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, float) or isinstance(b, float):
            return abs(float(a) - float(b)) <= tol
        return a == b
    for k in want:
        if k not in got or not same(got[k], want[k]):
            bad.append(k)
    return bad
