"""Array-native transport of a video's detections.

The reference moves detections as JSON protocol dicts or one .mat per frame
(utils/protocol.py:528-555: load_frame_to_det -> {frame: (boxes [B,4], zs [B,C])}, load_det_info ->
rows [frame, x1,y1,x2,y2, scores...]).  A config-2 video (300 x 10 000 x 200) is 2.4 GB of scores --
it cannot travel as dicts.  These helpers convert between the reference's containers and dense
``boxes [F,B,4]`` / ``scores [F,B,C]`` float32 arrays (ragged frames are padded with far-away 1x1
boxes whose scores are -inf), store / load them as one .npz, or -- the fast path -- as raw .npy files that
are memory-mapped, copied ONCE into pinned staging memory and uploaded on a private stream while the
previous video is processed (``VideoUploader``)."""
import numpy as np

def pad_boxes(n):
    """n filler boxes for ragged frames: 1x1 boxes far outside any image, 2 px apart (they overlap
    nothing, not even each other, so they add no edges to the suppression graph and the frame stays
    "regular").  Their scores are -inf: run NMS with ``score_thresh=-inf`` (score > thresh) to drop them."""
    x = -1.0e6 - 2.0 * np.arange(n, dtype=np.float32)
    return np.stack([x, np.full(n, -1.0e6, np.float32), x, np.full(n, -1.0e6, np.float32)], 1)



def arrays_from_frame_to_det(vid_proto, frame_to_det):
    """{frame_id: (boxes [B_f,4], zs [B_f,C])} -> (boxes [F,B,4] f32, scores [F,B,C] f32, counts [F]).
    Frames without detections get count 0."""
    frames = [f['frame'] for f in vid_proto['frames']]
    C = 0
    B = 0
    for fid in frames:
        if fid in frame_to_det and np.asarray(frame_to_det[fid][0]).size:
            B = max(B, np.asarray(frame_to_det[fid][0]).shape[0])
            C = max(C, np.asarray(frame_to_det[fid][1]).shape[1])
    boxes = np.tile(pad_boxes(max(B, 1)), (len(frames), 1, 1))
    scores = np.full((len(frames), max(B, 1), max(C, 1)), -np.inf, dtype=np.float32)
    counts = np.zeros(len(frames), dtype=np.int32)
    for i, fid in enumerate(frames):
        if fid not in frame_to_det:
            continue
        b, z = np.asarray(frame_to_det[fid][0]), np.asarray(frame_to_det[fid][1])
        if b.size == 0:
            continue
        n = b.shape[0]
        boxes[i, :n] = b.astype(np.float32)
        scores[i, :n, :z.shape[1]] = z.astype(np.float32)
        counts[i] = n
    return boxes, scores, counts


def frame_to_det_from_arrays(vid_proto, boxes, scores, counts=None):
    """Inverse of arrays_from_frame_to_det (frames with count 0 are omitted, like a missing .mat)."""
    out = {}
    for i, frame in enumerate(vid_proto['frames']):
        n = boxes.shape[1] if counts is None else int(counts[i])
        if n:
            out[frame['frame']] = (np.asarray(boxes[i, :n]), np.asarray(scores[i, :n]))
    return out


def det_info_from_arrays(vid_proto, boxes, scores, counts=None):
    """(boxes, scores) -> the reference's det_info rows [frame, x1,y1,x2,y2, scores...] (float64),
    utils/protocol.py:541-555."""
    rows = []
    for i, frame in enumerate(vid_proto['frames']):
        n = boxes.shape[1] if counts is None else int(counts[i])
        if n:
            rows.append(np.hstack([np.full((n, 1), frame['frame'], dtype=np.float64),
                                   np.asarray(boxes[i, :n], dtype=np.float64),
                                   np.asarray(scores[i, :n], dtype=np.float64)]))
    return np.vstack(rows) if rows else np.zeros((0, 5 + scores.shape[2]))


def save_video_npz(path, boxes, scores, counts=None):
    np.savez(path, boxes=np.asarray(boxes, dtype=np.float32), scores=np.asarray(scores, dtype=np.float32),
             counts=np.asarray(counts if counts is not None else np.full(len(boxes), boxes.shape[1]), dtype=np.int32))


def load_video_npz(path, device=None):
    """Load a video written by save_video_npz.  With ``device`` the arrays are staged through pinned
    host memory and uploaded asynchronously (returns torch tensors on that device)."""
    z = np.load(path if str(path).endswith('.npz') else str(path) + '.npz')
    boxes, scores, counts = z['boxes'], z['scores'], z['counts']
    if device is None:
        return boxes, scores, counts
    import torch
    tb = torch.from_numpy(boxes).pin_memory().to(device, non_blocking=True)
    ts = torch.from_numpy(scores).pin_memory().to(device, non_blocking=True)
    return tb, ts, torch.from_numpy(counts)


# ---------------------------------------------------------------------------------------------------
# Fast ingest: raw .npy files + memory map + ONE host pass into pinned memory
# ---------------------------------------------------------------------------------------------------

def save_video_raw(prefix, boxes, scores, counts=None):
    """One uncompressed .npy per array (``<prefix>.boxes.npy`` / ``.scores.npy`` / ``.counts.npy``): the
    loader memory-maps them, so the only host pass over a config-2 video's 2.4 GB of scores is the copy
    from the page cache into pinned memory (an .npz is first inflated into a pageable array)."""
    np.save(prefix + '.boxes.npy', np.ascontiguousarray(boxes, dtype=np.float32))
    np.save(prefix + '.scores.npy', np.ascontiguousarray(scores, dtype=np.float32))
    np.save(prefix + '.counts.npy', np.asarray(counts if counts is not None else np.full(len(boxes), boxes.shape[1]),
                                                dtype=np.int32))


def open_video_raw(prefix):
    """(boxes, scores, counts) as read-only memory maps of the files written by save_video_raw."""
    return (np.load(prefix + '.boxes.npy', mmap_mode='r'), np.load(prefix + '.scores.npy', mmap_mode='r'),
            np.load(prefix + '.counts.npy', mmap_mode='r'))


class VideoUploader(object):
    """Double-buffered file -> HBM pipeline: ``nbuf`` pinned staging slots and a private upload stream.
    ``submit(prefix)`` copies the memory-mapped arrays into the next free pinned slot (the one host
    pass) and enqueues the asynchronous H2D copy on the upload stream; it returns device tensors plus
    an event to wait for before the video is used (``VideoUploader.acquire(tb, ts, ev)`` makes the current
    stream wait and tells the allocator that this stream uses the tensors).
    While video k is processed, video k+1 is copied and uploaded -- the deployment loop of
    ``bench.py``'s ``upload_pipeline`` leg, fed from files."""

    def __init__(self, device, nbuf=2):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots = [None] * nbuf
        self.done = [None] * nbuf          # H2D of this slot finished -> the pinned memory may be rewritten
        self.k = 0

    def _slot(self, i, boxes, scores):
        torch = self.torch
        s = self.slots[i]
        if s is None or s[0].shape != tuple(boxes.shape) or s[1].shape != tuple(scores.shape):
            s = (torch.empty(tuple(boxes.shape), dtype=torch.float32).pin_memory(),
                 torch.empty(tuple(scores.shape), dtype=torch.float32).pin_memory())
            self.slots[i] = s
        return s

    def acquire(self, tb, ts, ev):
        """Make torch's current stream wait for the upload and register it as a user of the tensors."""
        cur = self.torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        tb.record_stream(cur)
        ts.record_stream(cur)

    def submit(self, prefix):
        torch = self.torch
        boxes, scores, counts = open_video_raw(prefix)
        i = self.k % len(self.slots)
        self.k += 1
        if self.done[i] is not None:
            self.done[i].synchronize()                       # the slot's previous upload has left host memory
        hb, hs = self._slot(i, boxes, scores)
        np.copyto(hb.numpy(), boxes)                         # page cache -> pinned: the only host pass
        np.copyto(hs.numpy(), scores)
        with torch.cuda.stream(self.stream):
            tb = hb.to(self.device, non_blocking=True)
            ts = hs.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.done[i] = ev
        return tb, ts, np.array(counts), ev
