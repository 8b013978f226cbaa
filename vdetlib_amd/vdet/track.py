"""Greedy tubelet generation of the reference's vdet/track.py: pick the top remaining detection as
anchor -> run the tracker plug-in -> suppress the detections the new track explains
(track_det_nms, on the GPU) -> repeat.

The reference's trackers (fcn_tracker :52-106, tld_tracker :18-49) are wrappers around external
MATLAB code and are out of scope; ``track_method(vid_proto, anchor_frame_id, anchor_bbox, opts)
-> [tracklet, ...]`` stays the plug-in boundary (tracklet = list of
{'frame','bbox','hash','score','anchor'}, utils/protocol.py:403-411).
"""
from collections import defaultdict

import numpy as np

from ..utils.cython_nms import track_det_nms, track_det_nms_batch     # noqa: F401 (track_det_nms: the reference's import)
from ..utils.log import logger as logging


def _external(name):
    def tracker(*args, **kwargs):
        raise RuntimeError("%s wraps an external MATLAB tracker (reference vdet/track.py) that is not part "
                           "of vdetlib_amd; pass your own track_method" % name)
    tracker.__name__ = name
    return tracker


fcn_tracker = _external('fcn_tracker')
tld_tracker = _external('tld_tracker')


def track_from_det(vid_proto, det_proto, track_method):
    """:109-119"""
    assert vid_proto['video'] == det_proto['video']
    tracks = []
    for idx, det in enumerate(det_proto['detections'], start=1):
        logging.info("tracking top No.{} in {}".format(idx, vid_proto['video']))
        tracks.extend(track_method(vid_proto, det))
    return {'video': vid_proto['video'], 'method': track_method.__name__, 'tracks': tracks}


def _nms_thres(opts):
    if hasattr(opts, 'nms_thres') and opts.nms_thres is not None:
        return opts.nms_thres
    return 0.3


def _run_tracker(track_method, vid_proto, anchor_frame_id, anchor_bbox, opts):
    """The reference retries once after restarting its MATLAB engine on ANY exception
    (:159-168, :225-234).  Here: if opts carries a callable ``on_tracker_error`` it is invoked and
    the tracker retried once; otherwise the exception propagates."""
    try:
        return track_method(vid_proto, anchor_frame_id, anchor_bbox, opts)
    except Exception:
        hook = getattr(opts, 'on_tracker_error', None) if hasattr(opts, 'on_tracker_error') else None
        if not callable(hook):
            raise
        hook(opts)
        return track_method(vid_proto, anchor_frame_id, anchor_bbox, opts)


def _prune_frame_dets(tracklets, rows_of_frame, alive, det_rows, nms_thres):
    """What a new track explains goes (reference :170-184 / :236-250): for every box of every new tracklet, the still-alive
    detections of that box's frame are put through ``track_det_nms`` against the box; whatever it does not return is dead.
    The reference makes one call per box; the boxes of a tracklet sit on different frames, so their problems are independent
    and run as ONE ``vdet_track_det_nms_batch`` call (one launch, one host wait).  A frame that comes up a second time (a
    tracker returning overlapping tracklets) depends on the first visit's result: the batch collected so far is flushed first,
    which keeps the reference's sequential order exactly.
    ``rows_of_frame``: frame id -> row numbers of that frame (ascending = descending score); ``alive`` is updated in place."""
    pending, in_batch = [], set()

    def flush():
        lives, track_rows = [], []
        for frame, bbox, rows in pending:
            live = rows[alive[rows]]
            if live.size:
                lives.append(live)
                track_rows.append([frame] + list(bbox))
        del pending[:]
        in_batch.clear()
        if not lives:
            return
        sizes = np.fromiter(map(len, lives), dtype=np.int64, count=len(lives))
        off = np.zeros(len(lives) + 1, dtype=np.int64)
        np.cumsum(sizes, out=off[1:])
        cat = np.concatenate(lives)
        keep, counts = track_det_nms_batch(np.asarray(track_rows, dtype=np.float32), det_rows[cat], off, nms_thres)
        start = np.repeat(off[:-1], sizes)                        # first row of every row's problem
        kept = (np.arange(cat.size) - start) < np.repeat(counts, sizes)     # entry q of problem k is a kept position iff q < counts[k]
        alive[cat] = False
        alive[cat[(keep[:cat.size] + start)[kept]]] = True

    for tracklet in tracklets:
        for box in tracklet:
            frame = box['frame']
            rows = rows_of_frame.get(frame)
            if rows is None:
                continue
            if frame in in_batch:
                flush()
            in_batch.add(frame)
            pending.append((frame, box['bbox'], rows))
    flush()


def _rows_by_frame(frame_keys):
    """{frame key: ascending row numbers of that frame} (the reference's ``frame_to_det_ids``, :203-205 / :137-139)."""
    keys = np.asarray(frame_keys)
    if keys.ndim == 1 and keys.size and keys.dtype.kind in 'iuf' and not (keys.dtype.kind == 'f' and np.isnan(keys).any()):
        uniq, inv = np.unique(keys, return_inverse=True)
        order = np.argsort(inv, kind='stable')
        bounds = np.searchsorted(inv[order], np.arange(len(uniq) + 1))
        # python numbers as keys: 5.0 == 5 and hash(5.0) == hash(5), so a tracklet's int frame id finds a float32 frame column
        return {uniq[k].item(): order[bounds[k]:bounds[k + 1]] for k in range(len(uniq))}
    positions = defaultdict(list)          # (frame ids that are not plain numbers: grouped the reference's way)
    for row, key in enumerate(frame_keys):
        positions[key].append(row)
    return {key: np.asarray(rows, dtype=np.int64) for key, rows in positions.items()}


def _greedy_loop(vid_proto, det_rows, frame_keys, score_of, anchor_of, track_method, opts):
    """The loop both entry points share (reference :140-186 / :207-252), on arrays.  ``det_rows``: float32 [N,6] rows
    (frame, x1,y1,x2,y2, score) already in descending score order.  State = one boolean per row; the scan position only
    moves forward (a row passed over is never an anchor again, like the reference's cursor); the loop ends when no alive
    row is left at or after it, when ``opts.max_tracks`` tracklets exist, or at the first anchor scoring below
    ``opts.thres``."""
    nms_thres = _nms_thres(opts)
    n = len(det_rows)
    rows_of_frame = _rows_by_frame(frame_keys)
    alive = np.ones(n, dtype=bool)
    scan = 0
    tracks = []
    while len(tracks) < opts.max_tracks:
        ahead = np.flatnonzero(alive[scan:])
        if ahead.size == 0:
            break
        anchor_row = scan + int(ahead[0])
        scan = anchor_row + 1
        if score_of(anchor_row) < opts.thres:
            logging.info("anchor below thres {}: {} tracks in total".format(opts.thres, len(tracks)))
            break
        logging.info("track {} of {}".format(len(tracks), vid_proto['video']))
        anchor_frame_id, anchor_bbox = anchor_of(anchor_row)
        new_tracks = _run_tracker(track_method, vid_proto, anchor_frame_id, anchor_bbox, opts)
        tracks.extend(new_tracks)
        _prune_frame_dets(new_tracks, rows_of_frame, alive, det_rows, nms_thres)
        logging.info("{} of {} detections left".format(int(alive.sum()), n))
    return tracks


def greedily_track_from_det(vid_proto, det_proto, track_method, score_fun, opts):
    """:122-186 -- detections as protocol dicts, ``score_fun(det)`` gives the class score."""
    assert vid_proto['video'] == det_proto['video']
    dets = sorted(det_proto['detections'], key=lambda x: score_fun(x), reverse=True)
    det_info = np.asarray([[det['frame'], ] + det['bbox'] + [score_fun(det), ] for det in dets],
                          dtype=np.float32).reshape(-1, 6)
    tracks = _greedy_loop(
        vid_proto, det_info, [det['frame'] for det in dets],
        lambda i: score_fun(dets[i]),
        lambda i: (dets[i]['frame'], [int(v) for v in dets[i]['bbox']]),
        track_method, opts)
    return {'video': vid_proto['video'], 'method': track_method.__name__, 'tracks': tracks}


def greedily_track_from_raw_dets(vid_proto, det_info, track_method, class_idx, opts):
    """:189-252 -- det_info [N, 5+C] rows (frame, x1,y1,x2,y2, class scores...); the class score is
    column 4+class_idx.  Rows are ordered by a STABLE descending sort of the float64 score, then
    cast to float32 (:200-201)."""
    det_info = np.asarray(det_info)
    sel = det_info[:, [0, 1, 2, 3, 4, 4 + class_idx]]
    order = np.argsort(-sel[:, 5], kind='stable')       # == sorted(..., key=score, reverse=True)
    sel = np.asarray(sel[order], dtype=np.float32)
    tracks = _greedy_loop(
        vid_proto, sel, sel[:, 0],
        lambda i: sel[i][-1],
        lambda i: (int(sel[i][0]), [int(v) for v in sel[i][1:5]]),
        track_method, opts)
    return {'video': vid_proto['video'], 'method': track_method.__name__, 'tracks': tracks}
