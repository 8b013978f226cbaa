#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL -- under the driver's torch.distributed.run line, or plain `python bench.py --gpus N`, which
     launches its N ranks itself: self_launch)

A "step" = one pass of the hot path over one synthetic video per GPU, inputs already resident in
HBM:
  NMS   per-(frame,class) greedy NMS of all boxes (vdet/image_det.py:117-123 over
        vdet/video_det.py:89-99 == utils/nms.pyx vid_nms per class),
  TEMP  the temporal pass over the [frame x box x class] score volume (vdet/tubelet_cls.py:386-414),
  LINK  greedy tubelet generation for every class (vdet/track.py:189-252, built-in IoU-linking
        tracker, track_det_nms suppression) + tubelet re-scoring: spatial max-pooling / box
        regression, completion, temporal max-pool (vdet/tubelet_cls.py:493-535, :284-303, :386-414),
and for N > 1 the RCCL all-gather of the per-video results (final tubelets + per-(frame,class) kept
counts).  Workload at N=1 = BASELINE.json configs[1] (+ configs[2]'s tracking on the same video):
300 frames x 10 000 boxes x 200 classes.  Videos are sharded one per GPU ("weak" scaling).

Prints ONE JSON line (rank 0) with BASELINE.json's metric plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "boxes/sec whole-node (NMS+temporal-conv+link), 300f\u00d710k-box synth; mAP parity"   # BASELINE.json


def drop_c_stdio():
    """RCCL prints a version banner at communicator creation with C stdio (this build: unconditionally; it sits in libc's
    buffer until exit when stdout is a pipe or a file).  The bench's stdout is ONE JSON line: python's own buffer goes out
    first, then libc's pending output is flushed into /dev/null."""
    import ctypes
    sys.stdout.flush()
    try:
        keep = os.dup(1)
        null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(null, 1)
        ctypes.CDLL(None).fflush(None)
        os.dup2(keep, 1)
        os.close(null); os.close(keep)
    except Exception:
        pass


def emit_line(result):
    drop_c_stdio()
    sys.stdout.write(json.dumps(result) + "\n")
    sys.stdout.flush()


def synth_video_cuda(torch, seed, F, B, C, device, kind="rand"):
    """boxes [F,B,4] (integer-valued f32, 1280x720, SURVEY 8d recipe), scores [F,B,C] f32 ~U(0,1) (softmax-like) or,
    kind="randn", ~N(0,1) (SVM-margin-like: both signs, many exponents -- another radix-digit distribution for the sort)."""
    g = torch.Generator(device=device).manual_seed(seed)
    x1 = torch.rand(F, B, generator=g, device=device) * 1230
    y1 = torch.rand(F, B, generator=g, device=device) * 670
    w = 10 + torch.rand(F, B, generator=g, device=device) * 290
    h = 10 + torch.rand(F, B, generator=g, device=device) * 290
    boxes = torch.stack([x1, y1, torch.clamp(x1 + w, max=1279), torch.clamp(y1 + h, max=719)], -1).round().contiguous()
    if kind == "randn" or os.environ.get("VDET_BENCH_SCORES") == "randn":
        scores = torch.randn(F, B, C, generator=g, device=device)
    else:
        scores = torch.rand(F, B, C, generator=g, device=device)
    return boxes, scores


def synth_video_reference_stream(torch, seed, F, B, C, device):
    """BASELINE.md section 3 / SURVEY 8(d): the HOST generator `np.random.RandomState(1000 * config + video)` -- exactly
    tests/synth.video(seed, F, B, C): integer boxes, scores ((rank + 0.5) / B as f32) tie-free per (frame, class), so the
    build's visiting order is the reference's (`scores.argsort()[::-1]` has no ties to break, utils/nms.pyx:25).  The uniform
    draws come from the host stream frame by frame; only the per-column ranking (argsort of argsort) runs on the device."""
    import numpy as np
    import synth
    rng = np.random.RandomState(seed)
    boxes = np.stack([synth.boxes_1(rng, B, False) for _ in range(F)], 0)
    scores = torch.empty((F, B, C), dtype=torch.float32, device=device)
    for f in range(F):
        r = torch.from_numpy(rng.rand(B, C)).to(device)                         # same stream as rng.rand(F, B, C)
        ranks = torch.argsort(torch.argsort(r, dim=0), dim=0)
        scores[f] = ((ranks.to(torch.float64) + 0.5) / B).to(torch.float32)
    return torch.from_numpy(boxes).to(device).contiguous(), scores


def synth_vid_batch(torch, dev, V=64, B=300, C=30, seed=777):
    """V VID-shaped synthetic videos concatenated along F: boxes [F,B,4], scores [F,B,C], frame offsets [V+1]"""
    import numpy as np
    g = torch.Generator(device=dev).manual_seed(seed)
    frames = torch.randint(400, 601, (V,), generator=g, device=dev).tolist()
    off = np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)
    Ft = int(off[-1])
    vid_of = torch.repeat_interleave(torch.arange(V, device=dev), torch.tensor(frames, device=dev))
    base = torch.rand(V, B, 4, generator=g, device=dev)
    x1, y1 = base[..., 0] * 1230, base[..., 1] * 670
    w, h = 10 + base[..., 2] * 290, 10 + base[..., 3] * 290
    bb = torch.stack([x1, y1, torch.clamp(x1 + w, max=1279), torch.clamp(y1 + h, max=719)], -1)          # [V,B,4]
    boxes = (bb[vid_of] + torch.randint(-3, 4, (Ft, B, 4), generator=g, device=dev)).round().contiguous()
    boxes[..., 2:] = torch.maximum(boxes[..., 2:], boxes[..., :2] + 4)
    sbase = torch.rand(V, B, C, generator=g, device=dev)
    scores = (0.8 * sbase[vid_of] + 0.2 * torch.rand(Ft, B, C, generator=g, device=dev)).contiguous()
    cnt = torch.randint(200, B + 1, (Ft,), generator=g, device=dev)
    padm = torch.arange(B, device=dev)[None, :] >= cnt[:, None]                                           # ragged frames
    px = -1.0e6 - 2.0 * torch.arange(B, device=dev, dtype=torch.float32)
    padb = torch.stack([px, torch.full_like(px, -1.0e6), px, torch.full_like(px, -1.0e6)], 1)
    boxes = torch.where(padm[..., None], padb[None], boxes).contiguous()
    scores = torch.where(padm[..., None], torch.full_like(scores, float("-inf")), scores).contiguous()
    return boxes, scores, off



def kept_dets_to_det_proto(video_name, kept_idx, kept_cnt, boxes, scores, class_names, topk):
    """One video's gathered NMS results -> a det_proto (utils/protocol.py:307-320 dict layout): a detection per kept
    (frame, box) with the scores of the classes it was kept for.  kept_idx [F,C,topk] / kept_cnt [F,C] numpy, boxes
    [F,B,4] / scores [F,B,C] numpy (the box protos a host has anyway)."""
    import hashlib
    dets = {}
    F, C = kept_cnt.shape
    for f in range(F):
        for c in range(C):
            for b in kept_idx[f, c, :min(int(kept_cnt[f, c]), topk)].tolist():
                d = dets.get((f, b))
                if d is None:
                    bb = [int(v) for v in boxes[f, b]]
                    d = dets[(f, b)] = {"frame": f + 1, "bbox": bb, "scores": [],
                                        "hash": hashlib.md5('{}_{}_{}_{}_{}_{}'.format(video_name, f + 1, *bb).encode()).hexdigest()}
                d["scores"].append({"class": class_names[c], "class_index": c + 1, "score": float(scores[f, b, c])})
    return {"video": video_name, "detections": [dets[k] for k in sorted(dets)]}


def sharded_core(args, torch, dist, vdist, dev, local, world, rank, V, steps, warmup, force_x, one_gpu, check_oracle=False):
    """BASELINE configs[3]: V videos sharded over the ranks (LPT by frames x boxes), `--streams` of them in flight per rank,
    and ONE exchange step per pass: ragged RCCL all-gathers of every video's results (tubelets with their re-scored boxes and
    pooled scores, the top-k kept detections per (frame, class) + counts).  A "step" = one pass over all the videos.  Rank 0
    turns one gathered video into det_proto / track_proto dicts and (check_oracle) checks it against the CPU oracle.
    Returns the result dict on rank 0 (None elsewhere)."""
    import numpy as np
    from vdetlib_amd import ops, _lib
    B, C, T, TOPK = args.boxes, args.classes, args.max_tracks, min(100, args.cap)
    TAPS = [0.25, 0.5, 0.25]
    rs = np.random.RandomState(3000)
    frames = [max(3, int(round(args.frames * (0.75 + 0.5 * u)))) for u in rs.rand(V)]      # videos differ in length
    owned = vdist.shard_lpt([f * B for f in frames], world)
    mine = owned[rank]
    vids = {v: synth_video_cuda(torch, 3000 + v, frames[v], B, C, dev, args.scores) for v in mine}
    nstreams = max(1, min(args.streams, max(len(mine), 1)))
    ctxs = [_lib.Context(local) for _ in range(nstreams)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    for cx in ctxs:
        cx.set_cache(True); cx.set_async(True)
    # the large per-video outputs (both temporal volumes, the kept lists) live in ONE set of buffers per stream, sized for the
    # rank's longest video: videos of different lengths would otherwise churn the caching allocator with multi-GB blocks of
    # ever new sizes next to 157 GB of resident inputs
    fmax = max([frames[v] for v in mine] or [1])
    arena = [(torch.empty(fmax * B * C, dtype=torch.float32, device=dev), torch.empty(fmax * B * C, dtype=torch.float32, device=dev),
              torch.empty(fmax * C * args.cap, dtype=torch.int32, device=dev)) for _ in range(nstreams)]

    def process(v, k):
        vb, vs = vids[v]
        cx = ctxs[k]
        with torch.cuda.stream(streams[k]):
            cx.invalidate()
            pooled, conv = ops.volume_pass(vs, args.window, TAPS, ctx=cx, out=arena[k][:2])
            ki, kc, tr, an, nt = ops.nms_track_volume(vb, vs, nms_thres=args.thresh, thres=args.track_thres, max_tracks=T,
                                                      link_thres=args.link_thres, cap=args.cap, sync=False, ctx=cx, pad=False,
                                                      keep_out=arena[k][2])
            det, tp, tb = ops.rescore_tracks(tr, nt, vb, vs, overlap_thres=args.pool_thres, window=args.window, sync=False, ctx=cx)
            Fv = frames[v]
            # the video's result records: tubelet rows (x1 y1 x2 y2 link-score | re-scored box | pooled score) and the top-k kept
            tub = torch.cat([tr.reshape(C * T * Fv, 5), tb.reshape(C * T * Fv, 4), tp.to(torch.float32).reshape(C * T * Fv, 1)], 1)
            kept = ki[:, :, :TOPK].reshape(Fv * C, TOPK).clone()      # (a COPY: the slice of the stream's reused buffer would be a view)
            return {"v": v, "tub": tub, "anchors": an.reshape(C * T, 3), "ntracks": nt, "kept": kept, "kcnt": kc.reshape(Fv * C)}

    exch_ms = []
    # The exchange: ROUNDS of one fixed-capacity all-gather (vdist.PackedExchange).  Round j carries every rank's j-th video; a
    # record = header + the video's five result arrays back to back, capacity = the longest video of the run, so a round needs no
    # count exchange and no host wait: it is enqueued on a side stream the moment the video's results exist and travels while the
    # rank's streams compute the next videos.  Only the last round of a pass can be exposed (reported: exposed_ms).
    do_x = world > 1 or force_x
    rounds = max(len(o) for o in owned)
    fcap = max(frames)
    FIELDS = (("tub", torch.float32, lambda Fv: C * T * Fv * 10), ("anchors", torch.float32, lambda Fv: C * T * 3), ("ntracks", torch.int32, lambda Fv: C),
              ("kept", torch.int32, lambda Fv: Fv * C * TOPK), ("kcnt", torch.int32, lambda Fv: Fv * C))
    nbytes = lambda Fv: [fn(Fv) * 4 for _, _, fn in FIELDS]
    xch = vdist.PackedExchange(sum((n + 15) // 16 * 16 for n in nbytes(fcap)), rounds, dev, force=force_x) if do_x else None
    if xch is not None:
        for j in range(rounds):
            if j < len(mine):
                xch.set_record(j, mine[j], frames[mine[j]], nbytes(frames[mine[j]]))
            else:
                xch.set_record(j, -1, 0, [])

    def one_pass(exchange=True):
        res = []
        for j in range(rounds):
            k = j % nstreams
            if j < len(mine):
                r = process(mine[j], k)
                res.append(r)
                if exchange and xch is not None:
                    with torch.cuda.stream(streams[k]):
                        xch.pack(j, [r["tub"], r["anchors"], r["ntracks"].to(torch.int32), r["kept"], r["kcnt"]])
            if exchange and xch is not None:
                ev = torch.cuda.Event()
                ev.record(streams[k])
                xch.launch(j, ev)
        for s_ in streams:
            torch.cuda.current_stream().wait_stream(s_)
        out = None
        if exchange and xch is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()          # every video of the pass is computed ...
            xch.join()
            e1.record()          # ... and every round has arrived: what lies between is the exchange NOT hidden behind compute
            exch_ms.append((e0, e1))
            out = {"bytes": sum(r[k].numel() * r[k].element_size() for r in res for k in ("tub", "anchors", "ntracks", "kept", "kcnt"))}
        return res, out

    def fence():
        if world > 1 or force_x:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 2)):
        one_pass()
        for cx in ctxs:
            try:
                cx.sync()
            except _lib.RetryError:
                pass
    fence()
    del exch_ms[:]
    t0 = time.perf_counter()
    for _ in range(steps):
        res, gathered = one_pass()
    fence()
    dt = time.perf_counter() - t0
    for cx in ctxs:
        cx.sync()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else dev)
    by_rank = [dt]
    if world > 1:
        allt = torch.empty(world, dtype=torch.float64, device=tmax.device)
        dist.all_gather_into_tensor(allt, tmax)
        by_rank = [float(x) for x in allt.tolist()]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    total_boxes = sum(frames) * B
    result = None
    if rank == 0:
        xinfo, protos, oracle_check = None, None, None
        if gathered is not None:
            xms = [a.elapsed_time(b) for a, b in exch_ms]
            heads = [[xch.record_of(j, r_)[0] for r_ in range(xch.world)] for j in range(rounds)]
            got = sorted(int(h[0]) for row in heads for h in row if h[0] >= 0)
            xinfo = {"exposed_ms": sum(xms) / len(xms), "exposed_ms_max": max(xms), "exchange_ms": sum(xms) / len(xms),
                     "payload_bytes_per_rank": gathered["bytes"], "padded_bytes_per_rank": rounds * xch.cap, "rounds": rounds,
                     "record_capacity_bytes": xch.cap, "backend": dist.get_backend(), "world": dist.get_world_size(),
                     "videos_gathered": len(got), "all_videos_present": got == list(range(V)),
                     "how": "one fixed-capacity all_gather_into_tensor per round (every rank's j-th video), enqueued on a side stream when the "
                            "video's results exist; exposed_ms = what is left after the pass's last video is computed"}
            # one video of the LAST rank's shard -> protocol dicts on rank 0 (boxes / scores of a remote video: regenerated
            # from its seed here; a deployment has the box protos on the host).  The shortest one: the oracle check below is
            # single-threaded python + C
            r_ = xch.world - 1
            while r_ > 0 and not owned[r_]:
                r_ -= 1
            slot = min(range(len(owned[r_])), key=lambda q: (frames[owned[r_][q]], q))
            hdr, fld = xch.record_of(slot, r_)
            v, Fv = int(hdr[0]), int(hdr[1])
            assert v == owned[r_][slot] and Fv == frames[v], (v, Fv, owned[r_][slot])
            tub = fld[0].view(torch.float32).reshape(C, T, Fv, 10)
            anc = fld[1].view(torch.float32).reshape(C, T, 3)
            ntr = fld[2].view(torch.int32)
            kept = fld[3].view(torch.int32).reshape(Fv, C, TOPK).cpu().numpy()
            kcnt = fld[4].view(torch.int32).reshape(Fv, C).cpu().numpy()
            c_best = int(torch.argmax(ntr).item())
            tp = ops.tracks_to_proto("synth_%d" % v, tub[c_best, :, :, :5].contiguous(), anc[c_best], int(ntr[c_best]))
            hb, hs = [t.cpu().numpy() for t in (vids[v] if v in vids else synth_video_cuda(torch, 3000 + v, Fv, B, C, dev, args.scores))]
            dp = kept_dets_to_det_proto("synth_%d" % v, kept[:min(Fv, 3)], kcnt[:min(Fv, 3)], hb, hs, ["c%d" % (c + 1) for c in range(C)], TOPK)
            protos = {"video": v, "from_rank": r_, "track_proto_class": c_best + 1, "tracks": len(tp["tracks"]),
                      "track_boxes": sum(len(t) for t in tp["tracks"]), "det_proto_frames": min(Fv, 3), "detections": len(dp["detections"])}
            if check_oracle:
                # the GATHERED records of that video against the CPU oracle: the kept lists of 4 frames x all classes
                # (utils/nms.pyx:17-68 per (frame, class)) and every tubelet of one class with its re-scoring
                # (vdet/track.py:189-252, vdet/tubelet_cls.py:493-535, :284-303, :386-414)
                from oracle import oracle
                oracle.build()
                fr = sorted(set([0, Fv // 3, (2 * Fv) // 3, Fv - 1]))
                nthr = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                widx, wcnt = oracle.nms_volume(hb[fr], np.ascontiguousarray(hs[fr]), args.thresh, cap=args.cap, threads=nthr)
                ok_nms = bool(np.array_equal(kcnt[fr], wcnt))
                for a, f in enumerate(fr):
                    for c in range(C):
                        n = min(int(wcnt[a, c]), TOPK)
                        ok_nms = ok_nms and bool(np.array_equal(kept[f, c, :n], widx[a, c, :n]))
                wt, wn, wpool, wbx = oracle.rescored_tubelets(hb, np.ascontiguousarray(hs[:, :, c_best:c_best + 1]), args.thresh, args.track_thres, T,
                                                              args.link_thres, args.pool_thres, args.window)
                n = int(wn[0])
                gt = tub[c_best].cpu().numpy()
                ok_tub = int(ntr[c_best]) == n and bool(np.array_equal(gt[:n, :, :5], wt[0, :n], equal_nan=True))
                has = ~np.isnan(wt[0, :n, :, 0])
                ok_tub = ok_tub and bool(np.array_equal(gt[:n, :, 5:9][has], wbx[0, :n][has]))
                ok_tub = ok_tub and bool(np.allclose(gt[:n, :, 9][has], wpool[0, :n][has].astype(np.float32), rtol=0, atol=1e-5))
                oracle_check = {"video": v, "frames": Fv, "nms_frames": fr, "nms_lists": len(fr) * C, "nms_ok": ok_nms,
                                "tubelet_class": c_best + 1, "tubelets": n, "tubelets_ok": bool(ok_tub)}
        bpb = 16 * C + 16
        loads8 = [sum(frames[v] for v in o) * B for o in vdist.shard_lpt([f * B for f in frames], 8)]
        result = {
            "metric": METRIC, "value": total_boxes * steps / dt, "unit": "boxes/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[3]: %d videos (%d..%d frames x %d boxes x %d classes) sharded over %d rank(s) by LPT, %d in flight per "
                                   "rank; per video NMS + temporal pass + %d tubelets/class + re-scoring; one ragged all-gather of all results "
                                   "per pass (a step = one pass over all videos)" % (V, min(frames), max(frames), B, C, world, nstreams, T),
                       "videos": V, "frames": frames, "boxes": B, "classes": C, "shards": owned, "parallelism": "video-sharded x%d" % world},
            "exchange": xinfo, "protocol_dicts": protos, "oracle_check": oracle_check,
            "ms_per_video": dt / steps / V * 1e3,
            "lpt_loads_world8_boxes": loads8,
            "lpt_imbalance_world8": max(loads8) / (sum(loads8) / 8.0),
            "per_rank": {"seconds": by_rank, "boxes": [sum(frames[v] for v in o) * B for o in owned],
                         "hbm_frac_algorithmic": [sum(frames[v] for v in o) * B * steps / max(t, 1e-9) * bpb / HBM_PEAK for o, t in zip(owned, by_rank)]},
            "roofline": {"bound": "hbm", "kernel": "whole path", "achieved": total_boxes * steps / dt * bpb / 1e9, "peak": world * HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": total_boxes * steps / dt * bpb / (world * HBM_PEAK), "traffic": None,
                         "algorithmic_bytes_per_box": bpb},
            "cpu_baseline": None,
        }
    for cx in ctxs:
        cx.close()
    del vids, ctxs, res, gathered, arena
    torch.cuda.empty_cache()
    return result


def run_sharded(args):
    """`python bench.py --videos N` (8 GPUs: `python bench.py --gpus 8 --videos 64`, which launches the ranks itself): BASELINE configs[3]"""
    import torch
    import torch.distributed as dist
    from vdetlib_amd import dist as vdist
    world, rank, local = vdist.env_world()
    one_gpu = os.environ.get("VDET_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_x = args.force_exchange
    vdist.init(backend=("gloo" if one_gpu else "nccl") if (world > 1 or force_x) else None, device=dev, force=force_x)
    result = sharded_core(args, torch, dist, vdist, dev, local, world, rank, args.videos, args.steps, args.warmup, force_x, one_gpu,
                          check_oracle=args.check_oracle)
    if world > 1 or force_x:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_line(result)
    else:
        drop_c_stdio()


def dropin_latency_leg(reps_small=200):
    """The boundary T-CNN actually calls: seconds per CALL of the three `utils.cython_nms` entry points through the python
    drop-in module (numpy in, python list out), beside the reference's own Cython module on one core of the build container
    (oracle/reference_ratio.json).  Calls of <= 640 rows are ONE launch (csrc/fused_kernels.hpp)."""
    import numpy as np
    import synth
    from vdetlib.utils import cython_nms

    def per_call(fn, reps):
        for _ in range(3):
            fn()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t) / reps * 1e3

    out = {"unit": "ms per call (python drop-in module, host numpy in -> python list out)",
           "reference_ms": {"nms_300": 1.0, "nms_1000": 4.8, "nms_2000": "13-15", "nms_10000": "107-162", "vid_nms_9000x30f": "~117 (3.5 s / 30 classes)",
                            "where": "BASELINE.md section 2: the reference's Cython module, one core of the build container"}}
    for n in (100, 300, 1000, 2000, 10000):
        d = synth.dets5(n, n)
        out["nms_%d" % n] = per_call(lambda: cython_nms.nms(d, 0.3), reps_small if n <= 1000 else 20)
    d6 = synth.dets6(77, 9000, 30)
    out["vid_nms_9000x30f"] = per_call(lambda: cython_nms.vid_nms(d6, 0.3), 20)
    for m in (300, 10000):
        dd = synth.dets6(78 + m, m, 1)
        tr = np.hstack([[1.0], dd[m // 2, 1:5] + 2.0]).astype(np.float32).reshape(1, 5)
        out["track_det_nms_%d" % m] = per_call(lambda: cython_nms.track_det_nms(tr, dd, 0.3), reps_small if m <= 1000 else 20)
    return out


def c1_dict_api_leg():
    """BASELINE configs[0] end to end through the reference's IMPORT NAMES (`vdetlib.*`, served by the build): the driver of
    tests/c1_flow.py -- the one the reference itself was run with (tests/golden/make_golden.py --c1-only) -- seconds per
    function beside the reference's on the same inputs, outputs compared with what the reference returned."""
    import gzip
    import c1_flow
    from vdetlib.vdet import video_det as V, image_det as I, track as K, tubelet_cls as T
    from vdetlib.utils import protocol as P, common as Cm
    mods = dict(V=V, I=I, K=K, T=T, P=P, Cm=Cm)
    inp = c1_flow.inputs()
    c1_flow.run(mods, inp, classes=[1, 2])           # warm-up (contexts, scratch)
    best, got = None, None
    for _ in range(2):
        sec, got = c1_flow.run(mods, inp)
        best = sec if best is None else {k: min(best[k], sec[k]) for k in sec}
    res = {"seconds": best, "shape": "%d frames x %d proposals x %d classes, %d tracks/class" % (c1_flow.F, c1_flow.B, c1_flow.C, c1_flow.MAX_TRACKS)}
    gp = os.path.join(ROOT, "tests", "golden", "c1_flow_golden.json.gz")
    if os.path.isfile(gp):
        with gzip.open(gp, "rt") as f:
            want = json.load(f)
        bad = c1_flow.compare(got, want, tol=1e-5)
        res["matches_reference_outputs"] = bad == []
        if bad:
            res["mismatching"] = bad
    rp = os.path.join(ROOT, "oracle", "reference_c1.json")
    if os.path.isfile(rp):
        ref = json.load(open(rp))
        res["reference_seconds"] = ref.get("seconds")
        res["reference_host"] = ref.get("host")
        if ref.get("seconds"):
            res["speedup_vs_reference"] = {k: ref["seconds"][k] / max(best.get(k, 0.0), 1e-9) for k in ref["seconds"] if k in best}
    return res


def vid_shape_leg(torch, ops, _lib, dev, taps, V=64, B=300, C=30, T=4):
    """64 VID-shaped synthetic videos (400-600 frames, 200-300 proposals per frame padded to 300 with far-away boxes and
    -inf scores like vdetlib_amd.io, 30 classes; proposals persist from frame to frame with a few pixels of jitter so
    that tubelets link): batched entry points vs one video at a time; results compared; one video vs the oracle."""
    import numpy as np
    boxes, scores, off = synth_vid_batch(torch, dev, V, B, C)
    Ft = int(off[-1])
    kw = dict(nms_thres=0.3, thres=0.5, max_tracks=T, link_thres=0.5)
    cb = _lib.Context(dev.index)
    cb.set_cache(True); cb.set_async(True)

    def batch():
        pooled, conv = ops.volume_pass(scores, 3, taps, ctx=cb, frame_off=off)
        return ops.video_batch(boxes, scores, off, cap=B, overlap_thres=0.7, window=3, sync=False, ctx=cb, pad=False, **kw), pooled

    def loop():
        outs = []
        for v in range(V):
            f0, f1 = int(off[v]), int(off[v + 1])
            vb, vs = boxes[f0:f1], scores[f0:f1]
            cb.invalidate()
            pooled, conv = ops.volume_pass(vs, 3, taps, ctx=cb)
            ki, kc, tr, an, nt = ops.nms_track_volume(vb, vs, cap=B, sync=False, ctx=cb, pad=False, **kw)
            det, tp, tb = ops.rescore_tracks(tr, nt, vb, vs, overlap_thres=0.7, window=3, sync=False, ctx=cb)
            outs.append((kc, tr, nt, tp))
        return outs

    res = {}
    for name, fn in (("batched", batch), ("one_video_at_a_time", loop)):
        for _ in range(2):
            cb.invalidate(); out = fn()
        cb.sync(); torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            cb.invalidate(); out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        cb.sync()
        res[name] = (dt, out)
    (tb_, (bo, bpool)), (tl_, lo) = res["batched"], res["one_video_at_a_time"]
    same = True
    for v in range(V):
        f0, f1 = int(off[v]), int(off[v + 1])
        kc, tr, nt, tp = lo[v]
        same = same and torch.equal(bo["keep_cnt"][f0:f1], kc) and torch.equal(bo["ntracks"][v], nt)
        same = same and torch.equal(bo["tracks"][v].nan_to_num(-7.0), tr.nan_to_num(-7.0)) and torch.equal(bo["pooled"][v].nan_to_num(-7.0), tp.nan_to_num(-7.0))
    # one video, two classes, against the oracle
    from oracle import oracle
    v = 5
    f0, f1 = int(off[v]), int(off[v + 1])
    hb, hs = boxes[f0:f1].cpu().numpy(), scores[f0:f1, :, :2].contiguous().cpu().numpy()
    wt, wn, wpool, wbx = oracle.rescored_tubelets(hb, hs, 0.3, 0.5, T, 0.5, 0.7, 3)
    ok = True
    for c in range(2):
        n = int(wn[c])
        ok = ok and int(bo["ntracks"][v, c]) == n and bool(np.array_equal(bo["tracks"][v][c, :n].cpu().numpy(), wt[c, :n], equal_nan=True))
        ok = ok and bool(np.allclose(bo["pooled"][v][c, :n].cpu().numpy(), wpool[c, :n], rtol=0, atol=1e-9, equal_nan=True))
    cb.close()
    nbox = Ft * B
    bytes_per_box = 16 * C + 16
    return {"videos": V, "frames_total": Ft, "boxes_per_frame": B, "classes": C, "tracks_per_class": T,
            "videos_per_s": V / tb_, "boxes_per_s": nbox / tb_, "ms_per_video": tb_ / V * 1e3,
            "hbm_frac_algorithmic": nbox / tb_ * bytes_per_box / HBM_PEAK, "algorithmic_bytes_per_box": bytes_per_box,
            "one_video_at_a_time": {"videos_per_s": V / tl_, "boxes_per_s": nbox / tl_, "ms_per_video": tl_ / V * 1e3},
            "speedup_vs_one_at_a_time": tl_ / tb_, "identical_to_one_at_a_time": bool(same), "oracle_sample_ok": bool(ok),
            "sample": "64 synthetic VID-shaped videos (400-600 frames x 200-300 of 300 proposals x 30 classes), NMS + temporal "
                      "max-pool / convolution + %d tubelets per class + re-scoring; oracle: video 5, classes 0-1" % T}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: re-execute under torch.distributed.run with N ranks on
    127.0.0.1 (one rank per GPU over RCCL) -- the same command line the driver would write.  Fewer than N visible devices is an
    error (exit 2), never a silent one-rank run.  Under torchrun the world size comes from the environment; a `--gpus` that
    disagrees with it is reported on stderr and the environment wins (the line's `n_gpus` is always the real world size)."""
    world_env = os.environ.get("WORLD_SIZE")
    # under torch.distributed.run every rank has WORLD_SIZE, RANK and LOCAL_RANK; a stray WORLD_SIZE=1 exported by a harness is not
    # a launcher (the ranks are then still ours to start)
    launched = world_env is not None and "LOCAL_RANK" in os.environ and "RANK" in os.environ
    if launched:
        if int(world_env) != args.gpus:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s: running %s ranks\n" % (args.gpus, world_env, world_env))
        return
    if args.gpus <= 1:
        return
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    if os.environ.get("VDET_BENCH_ONE_GPU") != "1":       # (test hook: every rank on device 0 over gloo)
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but %d HIP device(s) visible; not running fewer ranks than asked\n"
                             % (args.gpus, have))
            sys.exit(2)
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on this driver (RCCL peers)
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--boxes", type=int, default=10000)
    ap.add_argument("--classes", type=int, default=200)
    ap.add_argument("--cap", type=int, default=2048, help="survivor capacity per (frame,class)")
    ap.add_argument("--window", type=int, default=3)
    ap.add_argument("--thresh", type=float, default=0.3)
    ap.add_argument("--max-tracks", type=int, default=10)
    ap.add_argument("--track-thres", type=float, default=0.9, help="stop tracking below this score (90th percentile)")
    ap.add_argument("--link-thres", type=float, default=0.5)
    ap.add_argument("--pool-thres", type=float, default=0.7)
    ap.add_argument("--no-link", action="store_true", help="NMS + temporal only")
    ap.add_argument("--no-conv", action="store_true", help="skip the temporal convolution pass (temporal max-pool only)")
    ap.add_argument("--separate", action="store_true", help="vdet_nms_volume + vdet_track_volume instead of the fused call")
    ap.add_argument("--streams", type=int, default=4, help="videos in flight per GPU (one HIP stream + context each; round 3, with the tracking "
                                                           "loop in one launch: 4 (12.63 ms) > 3 (12.75) = 5 > 2 (14.3); round 2: 3 > 4 > 2)")
    ap.add_argument("--cpu-problems", type=int, default=1200, help="(frame,class) problems timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--scores", choices=["rand", "randn"], default="rand", help="synthetic score distribution")
    ap.add_argument("--separate-pass", action="store_true", help="temporal kernels + key transpose instead of the one volume pass")
    ap.add_argument("--sync-build", action="store_true", help="synchronous graph builds (no vdet_set_async)")
    ap.add_argument("--gate", choices=["heavy", "none"], default="none",
                    help="heavy: the throughput-bound phase (volume pass, graph, sort, walk) of step n+1 starts when that of step n is done "
                         "(HIP events between the streams), so exactly one video is in it while the latency-bound link chains of the "
                         "previous ones run underneath; none: streams run free")
    ap.add_argument("--max-frames", type=int, default=0, help="(ablation) tubelet length limit of the tracker (0 = the whole video)")
    ap.add_argument("--no-rescore", action="store_true", help="(ablation) tubelets without the spatial / temporal re-scoring")
    ap.add_argument("--videos", type=int, default=0, help="configs[3]: this many videos sharded over the ranks (LPT), one ragged exchange per pass; "
                    "0 (default): the headline step, one video per rank and step")
    ap.add_argument("--inputs", choices=["reference", "device"], default="reference",
                    help="reference: BASELINE.md section 3's host RandomState(1000 * config + video) stream, tie-free scores (default); "
                         "device: torch's device generator (rounds 1-4)")
    ap.add_argument("--check-oracle", action="store_true", help="--videos: check one gathered video against the CPU oracle on rank 0")
    ap.add_argument("--no-sharded-leg", action="store_true", help="skip the default line's configs[3] leg (64 videos sharded, one exchange)")
    ap.add_argument("--sharded-videos", type=int, default=64, help="videos of the default line's configs[3] leg")
    ap.add_argument("--profile", action="store_true", help="headline step + per-kernel timing only (rocprofv3 runs): no caveat legs")
    ap.add_argument("--no-coherent", action="store_true", help="skip the coherent-video leg (reported next to value, never part of it)")
    ap.add_argument("--no-upload", action="store_true", help="skip the PCIe-fed pipeline leg (reported next to value, never part of it)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the N > 1 exchange step -- process group, RCCL communicator, all-gather of device tensors from "
                         "every stream -- also in a world of ONE (what a single-GPU box can execute of configs[3])")
    args = ap.parse_args()
    self_launch(args)            # plain `python bench.py --gpus N` (no torchrun): becomes N ranks, one per GPU, over RCCL
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        args.no_cpu = True       # the CPU baseline / mAP-parity / PCIe legs are reported at N = 1 only
    if args.profile:
        args.no_cpu = args.no_coherent = args.no_upload = True
    if args.videos > 0:          # BASELINE configs[3]: many videos sharded over the ranks
        return run_sharded(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from vdetlib_amd import ops, _lib
    from vdetlib_amd import dist as vdist

    world, rank, local = vdist.env_world()
    # (test hook: VDET_BENCH_ONE_GPU=1 runs every rank on device 0 over gloo, to dry-run the N > 1 control
    #  flow on a single-GPU box; the real multi-GPU run is one rank per GPU over RCCL)
    one_gpu = os.environ.get("VDET_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_x = args.force_exchange and world == 1
    vdist.init(backend=("gloo" if one_gpu else "nccl") if (world > 1 or force_x) else None, device=dev, force=force_x)
    exch = {"events": [], "bytes": 0, "ok": True}      # per step: HIP events around the exchange on the step's stream
    F, B, C = args.frames, args.boxes, args.classes
    TOPK = 100
    TAPS = [0.25, 0.5, 0.25]     # the temporal convolution of the score volume (stand-in for the external TCN's first layer)

    # Consecutive videos are independent units of work: keep `--streams` of them in flight, each on its
    # own HIP stream with its own context (scratch buffers), so the latency-bound LINK kernels of one
    # video overlap the VALU-bound graph build / walk of the next.  Every stream works on its OWN video
    # (different boxes and scores).  All results of all K steps are complete before the timed region ends
    # (fence() synchronises the device).
    nstreams = max(1, args.streams)
    # inputs: BASELINE.md section 3's HOST generator, np.random.RandomState(1000 * config + video) -- integer boxes, scores tie-free
    # per (frame, class), so every timed list is visited in the reference's own order (its unstable argsort has no ties to
    # break); `--inputs device`: the fast torch device generator of rounds 1-4 (f32 U(0,1): ~3 tied pairs per 10 000-entry list)
    if args.inputs == "reference" and args.scores == "rand":
        vids = [synth_video_reference_stream(torch, 1000 * 2 + 100 * rank + k, F, B, C, dev) for k in range(nstreams)]
    else:
        vids = [synth_video_cuda(torch, 2000 + 100 * rank + k, F, B, C, dev, args.scores) for k in range(nstreams)]
    boxes, scores = vids[0]
    ctx = _lib.get_context(local)
    gathered = None
    ctxs = [ctx] + [_lib.Context(local) for _ in range(nstreams - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    for cx in ctxs:
        cx.set_cache(True)   # the volume pass, NMS and LINK of one step share keys, suppression graph and sorted lists
        cx.set_async(not args.sync_build)   # no host synchronisation inside a step once the first graph was built
    step_no = [0]
    last_out = [None] * nstreams     # every stream's most recent step outputs (kept alive: compared after the timed region)
    heavy_done = [None]      # event: the previous step's heavy phase has left the GPU
    upload_src = [None]      # --with-upload leg: pinned host videos fed over PCIe at the head of every step

    def step(exchange=True):
        nonlocal gathered
        k = step_no[0] % nstreams
        n = step_no[0]
        step_no[0] += 1
        cx = ctxs[k]
        vb, vs = vids[k]
        with torch.cuda.stream(streams[k]):
            gate = args.gate == "heavy" and nstreams > 1
            if gate and heavy_done[0] is not None:
                streams[k].wait_event(heavy_done[0])
            if upload_src[0] is not None:     # PCIe-fed pipeline: this stream's copy runs under the other streams' kernels
                hb, hs = upload_src[0][n % len(upload_src[0])]
                vb.copy_(hb, non_blocking=True)
                vs.copy_(hs, non_blocking=True)
            cx.invalidate()     # a new video: nothing may be reused from the previous step
            tub = None
            taps = None if (args.no_conv or args.window != len(TAPS)) else TAPS
            if not args.separate_pass:   # ONE read of the score volume: temporal max-pool + convolution + the sort keys
                pooled, conv = ops.volume_pass(vs, args.window, taps, ctx=cx)
            if args.no_link:
                keep_idx, keep_cnt = ops.nms_volume(vb, vs, args.thresh, cap=args.cap, sync=False, ctx=cx, pad=False)
            elif args.separate or gate:
                # the NMS call builds graph + sorted lists and walks them; the tracking call finds both in the context
                # (cache) and goes straight to the link loop -- the same kernels as the fused call, with a seam for the gate
                keep_idx, keep_cnt = ops.nms_volume(vb, vs, args.thresh, cap=args.cap, sync=False, ctx=cx, pad=False)
                if gate:
                    heavy_done[0] = torch.cuda.Event()
                    heavy_done[0].record(streams[k])
                tracks, anchors, ntracks = ops.track_volume(vb, vs, nms_thres=args.thresh, thres=args.track_thres,
                                                            max_tracks=args.max_tracks, link_thres=args.link_thres,
                                                            sync=False, ctx=cx)
            else:   # NMS survivors + tubelets from one call (same graph, same sorted lists)
                keep_idx, keep_cnt, tracks, anchors, ntracks = ops.nms_track_volume(
                    vb, vs, nms_thres=args.thresh, thres=args.track_thres, max_tracks=args.max_tracks,
                    link_thres=args.link_thres, max_frames=args.max_frames, cap=args.cap, sync=False, ctx=cx, pad=False)
            if args.separate_pass:
                if taps is None:
                    pooled = ops.temporal_maxpool(vs, args.window, ctx=cx)
                    conv = None if args.no_conv else ops.temporal_conv(vs, TAPS, bias=0.0, pad=0.0, ctx=cx)
                else:   # both temporal operators from one read of the volume
                    pooled, conv = ops.temporal_maxpool_conv(vs, args.window, TAPS, ctx=cx)
            elif conv is None and not args.no_conv:
                conv = ops.temporal_conv(vs, TAPS, bias=0.0, pad=0.0, ctx=cx)
            if not args.no_link and not args.no_rescore:
                det, tpool, tboxes = ops.rescore_tracks(tracks, ntracks, vb, vs, overlap_thres=args.pool_thres,
                                                        window=args.window, sync=False, ctx=cx)
                tub = (tracks, ntracks, tpool, tboxes)
            if (world > 1 or force_x) and exchange:   # the one exchange step: RCCL all-gather of the per-video results over xGMI
                # (same geometry on every rank: one fixed-shape collective per tensor, nothing the host
                # has to wait for -- a count exchange would stall the multi-video pipeline)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if tub is not None:
                    payload = torch.cat([tub[3].reshape(C, -1), tub[2].to(torch.float32).reshape(C, -1)], 1)
                    sent = (payload, keep_cnt)
                else:
                    sent = (keep_idx[:, :, :TOPK].contiguous(), torch.clamp(keep_cnt, max=TOPK))
                gathered = (vdist.all_gather_fixed(sent[0], force=force_x), vdist.all_gather_fixed(sent[1], force=force_x))
                e1.record()
                exch["events"].append((e0, e1))
                exch["bytes"] = sum(t.numel() * t.element_size() for t in sent)
                exch["last"] = (sent, gathered)
        last_out[k] = (keep_idx, keep_cnt, pooled, tub, conv)
        return last_out[k]

    def fence():
        if world > 1 or force_x:
            dist.barrier()
        torch.cuda.synchronize()

    def warm():
        """Untimed steps.  Returns True when an asynchronous graph build outgrew the scratch sized by the first video
        (vdet_sync: VDET_EAGAIN, the scratch has been enlarged) -- on ANY rank, so that all ranks repeat together."""
        for _ in range(max(args.warmup, nstreams)):
            step()
        again = 0
        for cx in ctxs:
            try:
                cx.sync()
            except _lib.RetryError:
                again = 1
        if world > 1:
            flag = torch.tensor([again], dtype=torch.int32, device="cpu" if one_gpu else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            again = int(flag.item())
        fence()
        return bool(again)

    # priming (untimed, before the W warm-up steps): two rounds over the streams, so that every context has sized its
    # scratch and torch's caching allocator holds a block for every output of every stream -- a first-touch hipMalloc
    # inside the timed region stalls all streams (seen once as a 2x step time right after a cold start)
    for _ in range(2 * nstreams):
        step()
    for cx in ctxs:
        try:
            cx.sync()
        except _lib.RetryError:
            pass
    fence()
    if warm():
        warm()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    for cx in ctxs:
        cx.sync()       # surfaces latched device-side failures (capacity / zero union)
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else dev)
    by_rank = None
    if world > 1:
        # every rank's own time for its K steps (between the same two barriers): the line reports them next to the MAX
        allt = torch.empty(world, dtype=torch.float64, device=tmax.device)
        dist.all_gather_into_tensor(allt, tmax)
        by_rank = [float(x) for x in allt.tolist()]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    boxes_per_s = world * F * B * args.steps / dt
    exchange = None
    if exch["events"]:
        ev = exch["events"][-args.steps:]                     # the timed steps' exchanges (everything has completed: fence())
        xms = [a.elapsed_time(b) for a, b in ev]
        sent, got = exch["last"]
        mine = all(torch.equal(g[rank].nan_to_num(-7.0), t_.nan_to_num(-7.0) if t_.dim() else t_[None].nan_to_num(-7.0))
                   for t_, g in zip(sent, got))               # this rank's slot of the gathered result is what it sent
        exchange = {"exchange_ms": sum(xms) / len(xms), "exchange_ms_max": max(xms),
                    "payload_bytes_per_rank": exch["bytes"], "gathered_bytes_per_rank": exch["bytes"] * max(world, 1),
                    "backend": dist.get_backend(), "world": dist.get_world_size(), "own_slot_matches": bool(mine),
                    "note": "HIP events on the step's stream around the two all_gather_into_tensor calls (tubelet boxes + pooled "
                            "scores, kept counts); the collectives run on RCCL's stream, ordered after the step's kernels"}

    # ---- the TIMED configuration's own results: the last timed step of every stream (asynchronous builds, `--streams` videos
    # in flight on their own streams and contexts) against a synchronous run of the same video on one fresh context, one
    # video at a time on the default stream -- and, below (cpu_baseline.parity_checked_other_video), a video other than
    # the first against the CPU oracle
    timed_check = None
    if rank == 0:
        torch.cuda.synchronize()
        timed = list(last_out)
        ref_ctx = _lib.Context(local)
        ref_ctx.set_cache(True)
        taps_ref = None if (args.no_conv or args.window != len(TAPS)) else TAPS

        def same(a, b):
            if a is None or b is None:
                return a is None and b is None
            if a.is_floating_point():
                return bool(torch.equal(a.nan_to_num(-7.0), b.nan_to_num(-7.0)))
            return bool(torch.equal(a, b))
        per_stream = []
        for k in range(nstreams):
            if timed[k] is None:
                continue
            vb, vs = vids[k]
            ref_ctx.invalidate()
            rp, rc = ops.volume_pass(vs, args.window, taps_ref, ctx=ref_ctx)
            if rc is None and not args.no_conv:
                rc = ops.temporal_conv(vs, TAPS, bias=0.0, pad=0.0, ctx=ref_ctx)
            ki, kc, pooled_t, tub_t, conv_t = timed[k]
            ok = same(pooled_t, rp) and same(conv_t, rc)
            if args.no_link:
                ri, rcnt = ops.nms_volume(vb, vs, args.thresh, cap=args.cap, ctx=ref_ctx, pad=False)
            else:
                ri, rcnt, rtr, ran, rnt = ops.nms_track_volume(vb, vs, nms_thres=args.thresh, thres=args.track_thres,
                                                               max_tracks=args.max_tracks, link_thres=args.link_thres,
                                                               max_frames=args.max_frames, cap=args.cap, ctx=ref_ctx, pad=False)
                if tub_t is not None:
                    rdet, rtp, rtb = ops.rescore_tracks(rtr, rnt, vb, vs, overlap_thres=args.pool_thres, window=args.window, ctx=ref_ctx)
                    ok = ok and same(tub_t[0], rtr) and same(tub_t[1], rnt) and same(tub_t[2], rtp) and same(tub_t[3], rtb)
                    del rdet, rtp, rtb
                del rtr, ran, rnt
            live = torch.arange(ki.shape[2], device=dev)[None, None, :] < kc[:, :, None]     # (pad=False: the tail is uninitialised)
            ok = ok and same(kc, rcnt) and bool(torch.equal(torch.where(live, ki, -1), torch.where(live, ri, -1)))
            per_stream.append(bool(ok))
            del ri, rcnt, rp, rc, live
        timed_check = {"timed_outputs_identical": bool(per_stream) and all(per_stream), "streams_checked": len(per_stream),
                       "against": "one fresh synchronous context, one video at a time on the default stream (every output tensor "
                                  "of the step: kept indices + counts, both temporal volumes, tubelets, re-scored tubelets)"}
        ref_ctx.sync()
        del ref_ctx, timed
        torch.cuda.empty_cache()

    # ---- per-kernel timing (HIP events on the kernels' stream), outside the timed region
    result = None
    if rank == 0:
        torch.cuda.synchronize()
        step_no[0] = 0
        ctx.set_timing(2)
        reps = 3
        for _ in range(reps):
            step_no[0] = 0          # per-kernel timing: one video at a time on stream 0
            out = step(exchange=False)    # (rank 0 only: no collective here); video 0: the CPU parity sample below checks it
            torch.cuda.synchronize()
        ctx.sync()
        agg = {k: [ms, n] for k, (ms, n) in ctx.last_timing().items()}
        ctx.set_timing(0)
        stages = {k: {"ms_per_step": v[0] / reps, "launches_per_step": v[1] // reps,
                      "avg_launch_ms": (v[0] / v[1]) if v[1] else 0.0} for k, v in agg.items() if v[1]}
        dom = max(stages, key=lambda k: stages[k]["ms_per_step"])
        bytes_per_box = 16 * C + 16                       # algorithmic bytes per box
        units_per_launch = F * B / stages[dom]["launches_per_step"]
        achieved = bytes_per_box * units_per_launch / (stages[dom]["avg_launch_ms"] * 1e-3)
        traffic = None
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(pj):
            try:
                traffic = json.load(open(pj)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # what bounds each stage (VALU ceiling measured by devtools/valu_bench, profiles/)
        BOUND = {"iou_bits": "valu (pair tests + list entries, graph_lists_kernel)", "adj_build": "latency (adj_finish_kernel: pads + records; the lists themselves come out of the iou_bits stage = graph_lists_kernel)", "sort": "valu issue + lds (equalised counting sort)", "walk": "valu 47% / salu 46% / lds 49% busy, no pipe saturated (profiles/r06_pmc_sq*.csv)", "temporal": "hbm",
                 "transpose_keys": "hbm", "track_link": "latency (serial chain)", "track_pick": "latency", "rescore_spatial": "l2/valu",
                 "rescore_series": "latency", "track_suppress": "valu-issue", "iou_bits_general": "valu", "other": "latency"}
        for k, v in stages.items():
            v["bound"] = BOUND.get(k, "")
            v["hbm_frac_algorithmic"] = bytes_per_box * F * B / (v["ms_per_step"] * 1e-3) / HBM_PEAK
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK, "traffic": traffic,
                    "algorithmic_bytes_per_box": bytes_per_box,
                    "whole_path_frac": boxes_per_s / world * bytes_per_box / HBM_PEAK,
                    "stages": stages}
        # the temporal kernel is the one genuinely HBM-bound stage: report its own stream rate too
        if "temporal" in stages:
            fused_t = (not args.no_conv) and args.window == len(TAPS)
            # bytes this stage really moves per element: read 4 + max-pool 4 (+ convolution 4) (+ the sort keys 4
            # when it is the one volume pass)
            tb = (8.0 + (4.0 if fused_t else 0.0) + (0.0 if args.separate_pass or C % 4 else 4.0)) * F * B * C
            roofline["temporal_GBps"] = tb / (stages["temporal"]["ms_per_step"] * 1e-3) / 1e9
            stages["temporal"]["hbm_frac_actual_bytes"] = roofline["temporal_GBps"] * 1e9 / HBM_PEAK

        cpu = None
        if not args.no_cpu:
            from oracle import oracle
            oracle.build()
            nprob = max(1, args.cpu_problems)
            nf = min(F, 6)
            nc = min(C, max(1, nprob // nf))
            hb = boxes[:nf].cpu().numpy()
            hs = scores[:nf, :, :nc].contiguous().cpu().numpy()
            t1 = time.perf_counter()
            widx, wcnt = oracle.nms_volume(hb, hs, args.thresh, cap=args.cap)
            oracle.temporal_maxpool(hs, args.window)
            wconv = None if args.no_conv else oracle.temporal_conv(hs, TAPS, 0.0, 0.0)
            cdt = time.perf_counter() - t1
            cpu_boxes = nf * B * (nc / C)
            t_nms_per_box = cdt / cpu_boxes                 # seconds per box (all C classes), NMS + TEMP
            sample = ("NMS+TEMP on %d frames x %d classes x %d boxes (%d nms problems + temporal max-pool) in %.1f s"
                      % (nf, nc, B, nf * nc, cdt))
            gi = out[0][:nf, :nc].cpu().numpy()
            gc = out[1][:nf, :nc].cpu().numpy()
            live = np.arange(gi.shape[2])[None, None, :] < gc[:, :, None]       # (the step leaves the padding uninitialised)
            parity = bool(np.array_equal(gc, wcnt) and np.array_equal(np.where(live, gi, -1), widx))
            if nf > 1:   # temporal ops: the sample's last frame sees padding instead of frame nf, so compare the frames before it
                wpool = oracle.temporal_maxpool(hs, args.window)
                parity = parity and bool(np.array_equal(out[2][:nf - 1, :, :nc].cpu().numpy(), wpool[:nf - 1]))
                if wconv is not None:
                    parity = parity and bool(np.allclose(out[4][:nf - 1, :, :nc].cpu().numpy(), wconv[:nf - 1], rtol=0, atol=1e-5))
            t_link_per_box = 0.0
            if not args.no_link:
                # LINK on a bounded sample: the first frames of the video, one class, same options
                fl = min(F, 30)
                sb, ss = boxes[:fl].contiguous(), scores[:fl, :, :1].contiguous()
                t2 = time.perf_counter()
                wt, wa, wn = oracle.greedy_track_volume(sb.cpu().numpy(), ss[:, :, 0].cpu().numpy(), args.thresh,
                                                        args.track_thres, args.max_tracks, args.link_thres, 0)
                ldt = time.perf_counter() - t2
                t_link_per_box = ldt / (fl * B * (1.0 / C))
                sample += "; LINK (greedy tubelets, %d tracks) on %d frames x 1 class x %d boxes in %.1f s" % (wn, fl, B, ldt)
                ctx.invalidate()
                gt, ga, gn = ops.track_volume(sb, ss, nms_thres=args.thresh, thres=args.track_thres,
                                              max_tracks=args.max_tracks, link_thres=args.link_thres)
                ctx.invalidate()
                parity = parity and int(gn[0]) == wn and bool(np.array_equal(gt[0, :wn].cpu().numpy(), wt[:wn], equal_nan=True))
            ref_ratio = None
            rj = os.path.join(ROOT, "oracle", "reference_ratio.json")
            if os.path.isfile(rj):       # C port vs the reference's Cython module, measured in the build container
                try:
                    ref_ratio = json.load(open(rj))
                except Exception:
                    ref_ratio = None
            cpu = {"value": 1.0 / (t_nms_per_box + t_link_per_box), "unit": "boxes/s", "cores": 1, "kind": "port",
                   "reference_ratio": ref_ratio,
                   "sample": sample + "; oracle/vdet_oracle.c + oracle/oracle.py, single thread; per-box times of the "
                                      "stages are added (value = boxes/s through the same stages as the GPU step)",
                   "nms_temp_boxes_per_s": 1.0 / t_nms_per_box, "parity_checked": parity}

        cpu_all = None
        if not args.no_cpu:
            # the same NMS sample on every host core (independent problems on threads; the reference
            # is single-threaded -- SURVEY 8d asks for both a 1-core and an all-core figure)
            nthr = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if nthr > 1:
                t4 = time.perf_counter()
                midx, mcnt = oracle.nms_volume(hb, hs, args.thresh, cap=args.cap, threads=nthr)
                oracle.temporal_maxpool(hs, args.window)
                if not args.no_conv:
                    oracle.temporal_conv(hs, TAPS, 0.0, 0.0)
                mdt = time.perf_counter() - t4
                same = [k for k in stages if not k.startswith(("track_", "rescore_"))]
                gpu_same_ms = sum(stages[k]["ms_per_step"] for k in same)
                cpu_all = {"value": cpu_boxes / mdt, "unit": "boxes/s", "cores": nthr, "kind": "port",
                           "gpu_same_stages_boxes_per_s": F * B / (gpu_same_ms * 1e-3),
                           "gpu_same_stages": "NMS+TEMP kernels of one video, one at a time (%s): %.2f ms" % ("+".join(sorted(same)), gpu_same_ms),
                           "sample": "NMS+TEMP only (no LINK), same %d nms problems on %d host threads in %.2f s"
                                     % (nf * nc, nthr, mdt),
                           "parity_checked": bool(np.array_equal(midx, widx) and np.array_equal(mcnt, wcnt))}

        map_par = None
        if not args.no_cpu and not args.no_link:
            # BASELINE "mAP parity", config 5 in miniature (the VID dataset is not available): VID-shaped
            # synthetic annotation protos with planted objects, tubelets + re-scoring on the GPU and
            # through the oracle, scored with the build's VOC-style evaluator
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
            import synth
            from vdetlib_amd import eval as vev
            gd, cd, annots = [], [], []
            mf, mb, mc, mt = 20, 120, 5, 4
            for seed in (51, 52, 53):
                vb, vs, annot = synth.vid_with_objects(seed, mf, mb, mc)
                annots.append(annot)
                tb_, ts_ = torch.from_numpy(vb).to(dev), torch.from_numpy(vs).to(dev)
                ctx.invalidate()
                tr_, an_, nt_ = ops.track_volume(tb_, ts_, nms_thres=0.3, thres=0.5, max_tracks=mt, link_thres=0.4)
                det_, pool_, ob_ = ops.rescore_tracks(tr_, nt_, tb_, ts_, overlap_thres=0.5, window=3)
                ctx.invalidate()
                gd += vev.detections_from_tracks(annot["video"], tr_.cpu().numpy(), nt_.cpu().numpy(), pool_.cpu().numpy(),
                                                 ob_.cpu().numpy())
                cd += vev.detections_from_tracks(annot["video"], *oracle.rescored_tubelets(vb, vs, 0.3, 0.5, mt, 0.4, 0.5, 3))
            gt = vev.ground_truth_from_annots(annots)
            aps_g, map_g = vev.evaluate(gd, gt)
            aps_c, map_c = vev.evaluate(cd, gt)
            map_par = {"gpu_mAP": map_g, "oracle_mAP": map_c, "identical_tubelets": bool(gd == cd),
                       "identical_AP": bool(aps_g == aps_c and map_g == map_c),
                       "sample": "3 synthetic VID-shaped videos x %d frames x %d proposals x %d classes with planted "
                                 "objects; tubelets (4 tracks/class) + spatial/temporal re-scoring; VOC-style AP@0.5" % (mf, mb, mc)}

        pcie = None
        if not args.no_cpu:
            # what the step would cost if the video had to come over PCIe first (never part of `value`)
            try:
                hp = torch.empty(scores.shape, dtype=torch.float32).pin_memory()
                dst = torch.empty_like(scores)
                best = None
                for _ in range(3):
                    torch.cuda.synchronize()
                    t3 = time.perf_counter()
                    dst.copy_(hp, non_blocking=True)
                    torch.cuda.synchronize()
                    d3 = time.perf_counter() - t3
                    best = d3 if best is None else min(best, d3)
                nbytes = scores.numel() * 4 + boxes.numel() * 4
                pcie = {"h2d_GBps": scores.numel() * 4 / best / 1e9, "upload_ms_per_video": nbytes / (scores.numel() * 4 / best) * 1e3,
                        "note": "pinned host -> HBM copy of one video's scores+boxes; overlappable with the previous video's step"}
                del hp, dst
            except Exception as e:       # a host without enough lockable memory: report, do not fail the bench
                pcie = {"error": str(e)[:200]}

        upload = None
        if not args.no_cpu and not args.no_upload and world == 1:
            # the PCIe-fed pipeline -- every step first uploads its video (pinned host -> HBM on the
            # step's own stream, so the copy of one video runs under the kernels of the others), then processes it.
            # Reported NEXT to `value` (which is measured with HBM-resident inputs), never instead of it.
            try:
                nsrc = 2
                src = []
                for i in range(nsrc):
                    hb = torch.empty(boxes.shape, dtype=torch.float32).pin_memory()
                    hs = torch.empty(scores.shape, dtype=torch.float32).pin_memory()
                    hb.copy_(vids[i % nstreams][0]); hs.copy_(vids[i % nstreams][1])
                    src.append((hb, hs))
                torch.cuda.synchronize()
                upload_src[0] = src
                step_no[0] = 0
                for _ in range(nstreams):
                    step(exchange=False)
                torch.cuda.synchronize()
                usteps = max(args.steps, 2 * nstreams)
                t5 = time.perf_counter()
                for _ in range(usteps):
                    step(exchange=False)
                torch.cuda.synchronize()
                udt = time.perf_counter() - t5
                for cx in ctxs:
                    cx.sync()
                upload_src[0] = None
                nbytes = (scores.numel() + boxes.numel()) * 4
                upload = {"boxes_per_s": F * B * usteps / udt, "ms_per_step": udt / usteps * 1e3, "steps": usteps,
                          "h2d_GBps_sustained": nbytes * usteps / udt / 1e9,
                          "pcie_bound_boxes_per_s": (pcie or {}).get("h2d_GBps", 0) * 1e9 / (nbytes / (F * B)) if pcie and "h2d_GBps" in pcie else None,
                          "note": "every step uploads its own video from pinned host memory (%d videos in flight, copies overlap "
                                  "the other streams' kernels), then runs the same step as `value`" % nstreams}
                del src
            except Exception as e:
                upload_src[0] = None
                upload = {"error": str(e)[:200]}

        # ---- caveats of `value`, measured here so that the line is self-contained (never part of `value`)
        torch.cuda.synchronize()
        n1 = max(6, nstreams * 2)
        for _ in range(2):
            step_no[0] = 0
            step(exchange=False)
        torch.cuda.synchronize()
        t6 = time.perf_counter()
        for _ in range(n1):
            step_no[0] = 0          # always stream 0 / context 0: one video at a time, nothing overlaps
            step(exchange=False)
        torch.cuda.synchronize()
        single_video_ms = (time.perf_counter() - t6) / n1 * 1e3
        ctx.sync()
        # how many of that last step's (frame, class) columns the equalised counting sort handed to the LSD kernel
        lists = {"lsd_fallback_lists": ctx.query(9), "lists": F * C}
        value_other = None
        if not args.no_cpu:
            # the other synthetic score distribution (another radix-digit / sub-bin pattern for the sort): same step
            other = "randn" if args.scores == "rand" else "rand"
            g = torch.Generator(device=dev).manual_seed(4242)
            for vb_, vs_ in vids:
                if other == "randn":
                    vs_.normal_(generator=g)
                else:
                    vs_.uniform_(generator=g)
            step_no[0] = 0
            for _ in range(2 * nstreams):
                step(exchange=False)
            torch.cuda.synchronize()
            no = max(args.steps // 2, 2 * nstreams)
            t7 = time.perf_counter()
            for _ in range(no):
                step(exchange=False)
            torch.cuda.synchronize()
            odt = time.perf_counter() - t7
            for cx in ctxs:
                cx.sync()
            value_other = {"scores": other, "value": F * B * no / odt, "ms_per_step": odt / no * 1e3, "steps": no}
        # A COHERENT video of the same size (what real proposals look like: utils/protocol.py:358-369 box protos of consecutive
        # frames): every proposal persists from frame to frame with a few pixels of jitter and keeps most of its score, so the
        # best detections of a class are the same few objects in every frame and the tubelets of a class are DISTINCT chains
        # through all frames (on independent frames the chains of all classes merge into a few).  Never part of `value`.
        # ---- the same step with RANDOM scores (torch U(0,1), ~3 tied pairs per list): the default inputs' scores are evenly spaced
        # ((rank + 0.5) / B per (frame, class)), which is the counting sort's best case -- every key alone in its bin; with random
        # scores about half of the keys share a bin and phase 6 of the sort has real work.  Never part of `value`.
        value_random = None
        if not args.no_coherent:
            g0 = torch.Generator(device=dev).manual_seed(555)
            for vb_, vs_ in vids:
                vs_.uniform_(generator=g0)
            step_no[0] = 0
            for _ in range(2 * nstreams):
                step(exchange=False)
            for cx in ctxs:
                try:
                    cx.sync()
                except _lib.RetryError:
                    pass
            torch.cuda.synchronize()
            no = max(args.steps // 2, 2 * nstreams)
            t8 = time.perf_counter()
            for _ in range(no):
                step(exchange=False)
            torch.cuda.synchronize()
            rdt = time.perf_counter() - t8
            for cx in ctxs:
                cx.sync()
            ctx.set_timing(2)
            for _ in range(2):
                step_no[0] = 0
                step(exchange=False)
                torch.cuda.synchronize()
            ctx.sync()
            rst = {k: round(ms / 2, 3) for k, (ms, n) in ctx.last_timing().items() if n}
            ctx.set_timing(0)
            value_random = {"value": F * B * no / rdt, "ms_per_step": rdt / no * 1e3, "steps": no, "stage_ms_one_video": rst,
                            "scores": "torch U(0,1) on the default boxes (ties possible; the sort's bins hold 0..6 keys)"}
        value_coherent = None
        if not args.no_coherent:
            g = torch.Generator(device=dev).manual_seed(977)
            for vb_, vs_ in vids:
                base = torch.rand(B, 4, generator=g, device=dev)
                x1, y1 = base[:, 0] * 1230, base[:, 1] * 670
                bb = torch.stack([x1, y1, torch.clamp(x1 + 10 + base[:, 2] * 290, max=1279), torch.clamp(y1 + 10 + base[:, 3] * 290, max=719)], -1)
                vb_.copy_((bb[None] + torch.randint(-3, 4, (F, B, 4), generator=g, device=dev)).round())
                vb_[..., 2:] = torch.maximum(vb_[..., 2:], vb_[..., :2] + 4)
                vs_.uniform_(generator=g).mul_(0.2).add_(0.8 * torch.rand(B, C, generator=g, device=dev)[None])
                del base, bb
            step_no[0] = 0
            for _ in range(2 * nstreams):
                step(exchange=False)
            for cx in ctxs:
                try:
                    cx.sync()
                except _lib.RetryError:
                    pass
            torch.cuda.synchronize()
            no = max(args.steps // 2, 2 * nstreams)
            t8 = time.perf_counter()
            for _ in range(no):
                step(exchange=False)
            torch.cuda.synchronize()
            cdt = time.perf_counter() - t8
            for cx in ctxs:
                cx.sync()
            ctx.set_timing(2)
            for _ in range(2):
                step_no[0] = 0
                step(exchange=False)
                torch.cuda.synchronize()
            ctx.sync()
            cst = {k: round(ms / 2, 3) for k, (ms, n) in ctx.last_timing().items() if n}
            ctx.set_timing(0)
            t9 = time.perf_counter()
            for _ in range(4):
                step_no[0] = 0
                step(exchange=False)
            torch.cuda.synchronize()
            value_coherent = {"value": F * B * no / cdt, "ms_per_step": cdt / no * 1e3, "steps": no,
                              "single_video_ms": (time.perf_counter() - t9) / 4 * 1e3, "stage_ms_one_video": cst,
                              "link_steps_memo_scanned": [ctx.query(4) + ctx.query(6), ctx.query(5) + ctx.query(7)],
                              "video": "proposals persist over the frames (jitter +-3 px), scores 0.8 * per-proposal + 0.2 * per-frame noise"}
        vid_shape = None
        if not args.no_cpu and world == 1:
            # BASELINE configs[4]'s SHAPE (ILSVRC-VID val: hundreds of frames x <= 300 ragged proposals x 30 classes; the
            # dataset itself is not available): 64 synthetic videos through the batched entry points (one graph / sort /
            # walk / temporal pass for all of them, tracking + re-scoring per video on its frame range) against the same
            # videos one at a time.  Never part of `value`.
            try:
                vid_shape = vid_shape_leg(torch, ops, _lib, dev, TAPS)
            except Exception as e:       # (report, do not fail the headline)
                vid_shape = {"error": repr(e)[:300]}
        # BASELINE configs[0]'s reference flow with per-class regressed boxes (fast_rcnn_det_vid's per-class loop + apply_image_nms
        # per (frame, class): vdet/video_det.py:89-99, vdet/image_det.py:117-123) on the device: vdet_det_nms_volume at c1 shape
        # (30 frames x 300 proposals x 30 classes + background, top-100) and at c2 shape.  Never part of `value`.
        c1_flow = None
        if not args.no_cpu and world == 1:
            try:
                def flow(Ff, Bf, Kf, reps):
                    g = torch.Generator(device=dev).manual_seed(31)
                    bx = torch.rand(Ff, Bf, 1, 4, generator=g, device=dev)
                    x1, y1 = bx[..., 0] * 1230, bx[..., 1] * 670
                    base = torch.stack([x1, y1, x1 + 10 + bx[..., 2] * 290, y1 + 10 + bx[..., 3] * 290], -1)
                    BX = (base + (torch.rand(Ff, Bf, Kf, 4, generator=g, device=dev) - 0.5) * 8).round().contiguous()
                    S = torch.rand(Ff, Bf, Kf, generator=g, device=dev)
                    fc = _lib.Context(local)
                    for _ in range(2):
                        ops.det_nms_volume(BX, S, score_thresh=0.05, topk=100, nms_thresh=args.thresh, ctx=fc, want_dets=True)
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for _ in range(reps):
                        out_ = ops.det_nms_volume(BX, S, score_thresh=0.05, topk=100, nms_thresh=args.thresh, ctx=fc, want_dets=True, sync=False)
                    fc.sync(); torch.cuda.synchronize()
                    ms = (time.perf_counter() - t) / reps * 1e3
                    fc.close()
                    return ms, int(out_[4].sum().item())
                ms1, kept1 = flow(30, 300, 31, 20)
                ms2, kept2 = flow(F, B, C + 1, 3)
                c1_flow = {"c1_ms": ms1, "c1_kept": kept1, "c1_problems": 30 * 30,
                           "reference_cpu_s": {"fast_rcnn_det_vid_threshold_topk": 0.06, "apply_image_nms_x900_top100": 0.19,
                                               "where": "BASELINE.md section 2: the reference's own code, 1 core of the build container "
                                                        "(not the GPU box's host)"},
                           "c2_ms": ms2, "c2_kept": kept2, "c2_problems": F * C,
                           "what": "threshold 0.05 -> best 100 -> per-class NMS of each class's OWN regressed boxes [F,B,K,4], rows + kept lists out"}
            except Exception as e:
                c1_flow = {"error": repr(e)[:300]}
        hbm_total = None
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(pj):
            try:
                hbm_total = json.load(open(pj)).get("_per_video", None)
            except Exception:
                hbm_total = None
        # ---- the boundary T-CNN calls: per-call latency of the drop-in module, and BASELINE configs[0] end to end through the
        # reference's import names, beside the reference's own seconds on the same inputs.  Never part of `value`.
        dropin, c1_dict = None, None
        if not args.no_cpu and world == 1:
            try:
                dropin = dropin_latency_leg()
            except Exception as e:
                dropin = {"error": repr(e)[:300]}
            try:
                c1_dict = c1_dict_api_leg()
            except Exception as e:
                c1_dict = {"error": repr(e)[:300]}
        # ---- the same step on the OTHER input generator (reference host stream <-> torch device generator), one video at a time,
        # with its NMS lists of two frames against the oracle.  Never part of `value`.
        ref_inputs = None
        if not args.no_cpu and world == 1:
            try:
                other_kind = "device" if (args.inputs == "reference" and args.scores == "rand") else "reference"
                rb, rsc = (synth_video_cuda(torch, 2000, F, B, C, dev, "rand") if other_kind == "device"
                           else synth_video_reference_stream(torch, 2000, F, B, C, dev))
                keep_v = vids[0]
                vids[0] = (rb, rsc)
                for _ in range(2):
                    step_no[0] = 0
                    o_ = step(exchange=False)
                torch.cuda.synchronize()
                tr0 = time.perf_counter()
                for _ in range(4):
                    step_no[0] = 0
                    o_ = step(exchange=False)
                torch.cuda.synchronize()
                rms = (time.perf_counter() - tr0) / 4 * 1e3
                ctx.sync()
                from oracle import oracle
                fr = [0, F - 1]
                widx, wcnt = oracle.nms_volume(rb[fr].cpu().numpy(), rsc[fr].contiguous().cpu().numpy(), args.thresh, cap=args.cap,
                                               threads=len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 1)
                gi, gc = o_[0][fr].cpu().numpy(), o_[1][fr].cpu().numpy()
                live = np.arange(gi.shape[2])[None, None, :] < gc[:, :, None]
                ref_inputs = {"inputs": other_kind, "single_video_ms": rms, "lsd_fallback_lists": ctx.query(9), "oracle_nms_lists": len(fr) * C,
                              "oracle_nms_ok": bool(np.array_equal(gc, wcnt) and np.array_equal(np.where(live, gi, -1), widx))}
                vids[0] = keep_v
                del rb, rsc, o_, keep_v
            except Exception as e:
                ref_inputs = {"error": repr(e)[:300]}
        # ---- BASELINE configs[3] at its written size on the GPU(s) at hand: 64 videos of c2 shape (0.75-1.25 x 300 frames) sharded by
        # LPT, `--streams` in flight, ONE ragged exchange per pass over RCCL (a world of one here; 8 ranks: the same code under
        # torchrun), a gathered video turned into protocol dicts and checked against the oracle.  Never part of `value`.
        sharded = None
        if not args.no_cpu and not args.no_sharded_leg and world == 1 and args.sharded_videos > 0:
            try:
                for cx in ctxs[1:]:
                    cx.close()
                vids[:] = []
                last_out[:] = [None] * nstreams
                del boxes, scores, out
                torch.cuda.empty_cache()
                made_pg = False
                if not dist.is_initialized():
                    vdist.init(backend="nccl", device=dev, force=True)
                    made_pg = True
                sharded = sharded_core(args, torch, dist, vdist, dev, local, 1, 0, args.sharded_videos, 2, 2, True, False, check_oracle=True)
                if sharded is not None:
                    sharded = {k: sharded[k] for k in ("value", "ms_per_step", "ms_per_video", "steps", "config", "exchange", "protocol_dicts",
                                                       "oracle_check", "lpt_loads_world8_boxes", "lpt_imbalance_world8", "roofline")}
                    sharded["config"] = {k: v for k, v in sharded["config"].items() if k not in ("frames", "shards")}
                    sharded["how_to_run_on_8_gpus"] = "python bench.py --gpus 8 --videos 64   (launches 8 ranks by itself; or torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench.py --gpus 8 --videos 64)"
                if made_pg:
                    dist.barrier()
                    dist.destroy_process_group()
            except Exception as e:
                sharded = {"error": repr(e)[:300]}

        result = {
            "metric": METRIC,
            "value": boxes_per_s, "unit": "boxes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "unpinned_by_nature": "the linking RULE (the reference's trackers are external MATLAB code: the built-in f32 IoU link "
                                  "stands in), the temporal-convolution taps (external Caffe TCN: a fixed 3-tap filter), the order of "
                                  "equal scores inside a frame (the reference's argsort is unstable); everything else is pinned to "
                                  "outputs of the reference (tests/golden)",
            "config": {"workload": ("configs[1]%s: 1 video/GPU, %d frames x %d boxes x %d classes; per-(frame,class) "
                                   "NMS thresh %.2f + temporal max-pool w=%d" + ("" if args.no_conv else " + 3-tap temporal convolution") + "%s%s") %
                                   ("" if args.no_link else "+[2]", F, B, C, args.thresh, args.window,
                                    "" if args.no_link else "; greedy tubelets: %d tracks/class (stop < %.2f), IoU-link "
                                    ">= %.2f, spatial max-pool IoU > %.2f + completion + temporal max-pool" %
                                    (args.max_tracks, args.track_thres, args.link_thres, args.pool_thres),
                                    "; RCCL all-gather of the final tubelets + kept counts" if (world > 1 or force_x) else ""),
                       "frames": F, "boxes": B, "classes": C, "scores": args.scores,
                       "parallelism": "video-per-gpu x%d, %d distinct videos in flight per GPU%s" % (
                           world, nstreams, ", heavy phases gated one at a time" if args.gate == "heavy" and nstreams > 1 else "")},
            "exchange": exchange,
            "per_rank": None if by_rank is None else {
                "ms_per_step": [t / args.steps * 1e3 for t in by_rank],
                "boxes_per_s": [F * B * args.steps / t for t in by_rank],
                "hbm_frac_algorithmic": [F * B * args.steps / t * (16 * C + 16) / HBM_PEAK for t in by_rank],
                "note": "each rank's own wall time for the K timed steps (its video per step + the exchange); value uses the MAX"},
            "single_video_ms": single_video_ms,          # one video at a time (no videos in flight): the latency of one step
            "value_other_scores": value_other,           # the same step on the other synthetic score distribution
            "value_random_scores": value_random,         # ... with random instead of evenly spaced scores (the sort has real bins to order)
            "value_coherent": value_coherent,            # ... on a coherent video (proposals persist from frame to frame)
            "hbm_traffic_per_video": hbm_total,          # sum of the PMC table (profiles/pmc_traffic.json), all kernels of one step
            "vid_shape": vid_shape,
            "lists": lists,
            "c1_reference_flow": c1_flow,
            "dropin_latency": dropin,                    # per-call latency of utils.cython_nms (the boundary T-CNN calls)
            "c1_dict_api": c1_dict,                      # configs[0] end to end through `vdetlib.*` beside the reference's seconds
            "value_other_inputs": ref_inputs,            # the step on the other input generator (device <-> reference host stream)
            "sharded64": sharded,                        # configs[3] at its written size (one exchange per pass, oracle-checked)
            "timed_check": timed_check,
            "inputs": ("BASELINE.md section 3 host generator: np.random.RandomState(1000 * 2 + video), integer boxes, scores tie-free per "
                       "(frame, class)" if (args.inputs == "reference" and args.scores == "rand") else "torch device generator, scores " + args.scores) +
                      "; HBM-resident when the timed region starts (the PCIe-fed rate is upload_pipeline.boxes_per_s, never `value`)",
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_allcores": cpu_all, "map_parity": map_par, "pcie": pcie,
            "upload_pipeline": upload,
        }
    if world > 1 or force_x:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_line(result)
    else:
        drop_c_stdio()


if __name__ == "__main__":
    main()
