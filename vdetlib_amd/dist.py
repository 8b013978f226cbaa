"""Multi-GPU execution of the hot path: one process per GPU, videos sharded across ranks, results
combined by ONE exchange step -- an RCCL all-gather over xGMI (torch.distributed backend "nccl" is
RCCL on ROCm; "gloo" runs the same code on CPU tensors for the tests).

The reference has no distributed code at all; every reference function
takes one ``vid_proto``, so videos are independent units: no data-path collective, only the final
gather of the per-video results (counts first, then a fixed-capacity padded payload).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None, force=False):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process, unless ``force``:
    a world of ONE still gets its process group -- and with backend "nccl" its RCCL communicator -- so that the
    exchange step can be executed on a single-GPU box exactly as it runs on eight)."""
    world, rank, local = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return world, rank, local


def shard_round_robin(n_items, rank, world):
    """Indices of the items (videos) this rank owns."""
    return list(range(rank, n_items, world))


def shard_lpt(costs, world):
    """Longest-processing-time-first assignment (cost = frames x boxes of a video); deterministic.
    Returns a list of index lists, one per rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world
    owned = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owned[r].append(i)
        load[r] += costs[i]
    return [sorted(o) for o in owned]


def class_slice(n_classes, rank, world):
    """Secondary partition for ONE huge video on several GPUs: shard by CLASS -- never by
    frame, so temporal windows and tubelets need no halo.  Every rank keeps the boxes (48 MB at c2) and
    a contiguous slice of the class axis of the score volume; the suppression graph is rebuilt on every
    rank (class-independent, a few ms), everything per (frame, class) splits.  Returns (c0, c1)."""
    per, rem = divmod(n_classes, world)
    c0 = rank * per + min(rank, rem)
    return c0, c0 + per + (1 if rank < rem else 0)


def _collective(group, force):
    """Is there a collective to run?  (a world of one only when the caller insists: ``force``)"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)


def all_gather_ragged(t, group=None, force=False):
    """All-gather tensors whose first dimension differs per rank: counts first, then one padded
    fixed-capacity payload (a single large collective instead of many small ones -- xGMI rings are
    per-link bound).  Returns the list of per-rank tensors (on every rank)."""
    if not _collective(group, force):
        return [t]
    if _stage_on_host(t, group):
        return [p.to(t.device) for p in all_gather_ragged(t.cpu(), group, force)]
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.tolist()
    cap = max(max(counts), 1)
    pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = torch.empty((world * cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    return [out[r * cap:r * cap + counts[r]] for r in range(world)]


def _stage_on_host(t, group):
    """gloo moves host memory: device tensors are staged through the CPU (tests / single-GPU dry runs
    of the N > 1 control flow; RCCL takes device pointers directly)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_fixed(t, group=None, force=False):
    """All-gather of same-shape tensors (every rank holds one video of the same geometry): ONE
    collective, no count exchange and therefore no host synchronisation -- the call only enqueues, so
    a rank can keep several videos in flight on different streams.  Returns [world, *t.shape]."""
    if not _collective(group, force):
        return t[None]
    world = dist.get_world_size(group)
    t = t.contiguous()
    if t.dim() == 0:
        t = t[None]
    if _stage_on_host(t, group):
        return all_gather_fixed(t.cpu(), group, force).to(t.device)
    # concatenated layout along dim 0 (accepted by both RCCL and gloo), viewed as [world, ...]
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.view((world,) + tuple(t.shape))


def gather_video_results(video_ids, keep_idx, keep_cnt, group=None, force=False):
    """Combine per-video NMS results across ranks.

    video_ids: list of the global ids of this rank's videos; keep_idx [V,F,C,K] int32 and
    keep_cnt [V,F,C] int32 for those videos (same F,C,K on all ranks).  Returns
    {video_id: (keep_idx [F,C,K], keep_cnt [F,C])} for ALL videos, on every rank."""
    ids = torch.tensor(video_ids, dtype=torch.int64, device=keep_idx.device).reshape(-1)
    g_ids = all_gather_ragged(ids, group, force)
    g_idx = all_gather_ragged(keep_idx, group, force)
    g_cnt = all_gather_ragged(keep_cnt, group, force)
    out = {}
    for r in range(len(g_ids)):
        for k, vid in enumerate(g_ids[r].tolist()):
            out[int(vid)] = (g_idx[r][k], g_cnt[r][k])
    return out


class PackedExchange(object):
    """The result exchange of BASELINE configs[3] as it should run on a node: ROUNDS of one fixed-capacity all-gather.

    Round j carries every rank's j-th video (ranks with fewer videos send an empty record).  A record = a small int64 header
    (video id, frames, element count of every field) + the fields back to back in ONE uint8 buffer of ``capacity`` bytes, the same
    on every rank -- so a round is a single ``all_gather_into_tensor`` with no count exchange and therefore NO host
    synchronisation: it is enqueued on a side stream as soon as the video's results exist (an event) and runs over xGMI while
    the rank's other streams compute the next videos.  Only the last round of a pass can be exposed.  ``all_gather_ragged``
    above stays the general-purpose form (counts first, one padded payload, host-synchronous).

    CPU tensors (gloo tests) take the same path without streams."""

    HDR = 16          # int64 words of header: [0] video id (-1: empty record), [1] frames, [2] n fields, [3 ..] bytes per field

    def __init__(self, capacity, rounds, device, group=None, force=False):
        self.cap = (int(capacity) + 8 * self.HDR + 15) // 16 * 16
        self.rounds, self.device, self.group, self.force = int(rounds), torch.device(device), group, force
        self.on = _collective(group, force)
        self.world = dist.get_world_size(group) if self.on else 1
        self.cuda = self.device.type == "cuda"
        self.send = [torch.zeros(self.cap, dtype=torch.uint8, device=self.device) for _ in range(self.rounds)]
        self.recv = [torch.zeros(self.world * self.cap, dtype=torch.uint8, device=self.device) for _ in range(self.rounds)] if self.on else None
        self.comm = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.done = [None] * self.rounds          # event: round j's collective of the previous pass has left send[j] / recv[j]
        self.gloo = self.on and dist.get_backend(group) == "gloo"

    def set_record(self, j, video, frames, field_nbytes):
        """Once per geometry: the header of this rank's record of round j (``video`` = -1 and no fields: an empty record).  The
        header is written here, on the host's time, so that ``pack`` only enqueues device-to-device copies."""
        if len(field_nbytes) > self.HDR - 3:
            raise ValueError("too many fields for the record header")
        off = 8 * self.HDR
        for n in field_nbytes:
            off = (off + int(n) + 15) // 16 * 16
        if off > self.cap:
            raise ValueError("record exceeds the exchange capacity (%d > %d bytes)" % (off, self.cap))
        hdr = [int(video), int(frames), len(field_nbytes)] + [int(n) for n in field_nbytes] + [0] * (self.HDR - 3 - len(field_nbytes))
        self.send[j][:8 * self.HDR].view(torch.int64).copy_(torch.tensor(hdr, dtype=torch.int64))
        if self.cuda:
            torch.cuda.synchronize(self.device)

    def pack(self, j, tensors):
        """The fields of round j's record into its send buffer, on the CURRENT stream (after the previous pass's collective on
        that buffer): device-to-device copies only.  Sizes must match ``set_record``."""
        if self.cuda and self.done[j] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.done[j])
        buf = self.send[j]
        off = 8 * self.HDR
        for t in tensors:
            n = t.numel() * t.element_size()
            buf[off:off + n].view(t.dtype).copy_(t.reshape(-1), non_blocking=True)
            off = (off + n + 15) // 16 * 16
        return off

    def launch(self, j, after=None):
        """Enqueue round j's all-gather on the side stream, after ``after`` (an event recorded where the record was packed)."""
        if not self.on:
            return
        if self.gloo and self.cuda:            # (single-GPU dry runs: gloo moves host memory)
            if after is not None:
                after.synchronize()
            out = torch.empty(self.world * self.cap, dtype=torch.uint8)
            dist.all_gather_into_tensor(out, self.send[j].cpu(), group=self.group)
            self.recv[j].copy_(out)
            return
        if not self.cuda:
            dist.all_gather_into_tensor(self.recv[j], self.send[j], group=self.group)
            return
        if after is not None:
            self.comm.wait_event(after)
        with torch.cuda.stream(self.comm):
            dist.all_gather_into_tensor(self.recv[j], self.send[j], group=self.group)
            ev = torch.cuda.Event()
            ev.record(self.comm)
        self.done[j] = ev

    def join(self):
        """Make the current stream wait for every collective enqueued so far (no host wait)."""
        if self.on and self.cuda and not self.gloo:
            torch.cuda.current_stream(self.device).wait_stream(self.comm)

    def record_of(self, j, r):
        """Rank r's record of round j: (header list, {k: field k as a flat uint8 view}); host-synchronous (reads the header).
        header = [video id (-1: empty), frames, n fields, bytes of field 0, bytes of field 1, ...]."""
        buf = (self.recv[j][r * self.cap:(r + 1) * self.cap] if self.on else self.send[j])
        hdr = buf[:8 * self.HDR].view(torch.int64).tolist()
        fields, off = [], 8 * self.HDR
        for k in range(int(hdr[2])):
            n = int(hdr[3 + k])
            fields.append(buf[off:off + n])
            off = (off + n + 15) // 16 * 16
        return hdr, fields
