"""-m gpu: BASELINE configs[3] (videos sharded across GPUs + one exchange step) as far as ONE GPU can take it:
(a) bench.py's N = 2 control flow end to end -- two ranks launched by torch.distributed.run, both on device 0,
    exchange over gloo (VDET_BENCH_ONE_GPU=1): barriers, the all-gather inside the timed step, MAX over ranks;
(b) eight "virtual ranks" in one process: shard_round_robin / shard_lpt + per-video results + the gather
    reproduce the serial result for 11 videos of different sizes.
(c) the RCCL transport itself, as far as one GPU goes: a world of ONE rank with backend "nccl" -- process group with
    device_id, RCCL communicator, all_gather_into_tensor of device tensors issued from bench.py's three streams inside
    the timed step, and vdetlib_amd.dist's ragged gather on device tensors.  Eight ranks over xGMI need the driver's
    multi-GPU node."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu_dry_run():
    env = dict(os.environ, VDET_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--frames", "12", "--boxes", "2000", "--classes", "16", "--no-cpu"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]           # rank 0 prints the one JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak"
    assert r["value"] > 0 and abs(r["value"] - 2 * 12 * 2000 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 1e-6
    assert "all-gather" in r["config"]["workload"]
    assert len(r["per_rank"]["ms_per_step"]) == 2 and max(r["per_rank"]["ms_per_step"]) == pytest.approx(r["ms_per_step"], rel=1e-6)
    assert r["exchange"]["world"] == 2 and r["exchange"]["exchange_ms"] > 0


def test_plain_python_bench_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` WITHOUT torchrun (the driver's scaling command line may be exactly this): bench.py re-executes
    itself under torch.distributed.run with two ranks; the line says n_gpus == 2 and carries a two-rank exchange.  Both for the
    headline step and for configs[3]'s `--videos` form.  (VDET_BENCH_ONE_GPU=1: both ranks on device 0 over gloo.)"""
    env = dict(os.environ, VDET_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
            "--frames", "12", "--boxes", "2000", "--classes", "16", "--no-cpu"]
    for extra, scaling in (([], "weak"), (["--videos", "4"], "strong")):
        p = subprocess.run(base + extra, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        r = json.loads(lines[0])
        assert r["n_gpus"] == 2 and r["exchange"]["world"] == 2 and r["scaling"] == scaling


def test_plain_python_bench_refuses_more_gpus_than_visible():
    """--gpus N with fewer than N devices: exit code 2 and a message, never a silent one-rank run."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VDET_BENCH_ONE_GPU")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "visible" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_rccl_exchange_executes_in_a_world_of_one():
    """bench.py --force-exchange under torch.distributed.run with ONE rank: init_process_group("nccl", device_id=...),
    the fixed-shape all-gathers of the tubelet payload and the kept counts from every stream in flight, inside the timed
    step; the gathered slot equals what was sent; the line carries the exchange time and bytes."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "3",
           "--frames", "12", "--boxes", "2000", "--classes", "16", "--no-cpu", "--force-exchange"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    x = r["exchange"]
    assert x["backend"] == "nccl" and x["world"] == 1 and x["own_slot_matches"] is True
    assert x["exchange_ms"] > 0 and x["payload_bytes_per_rank"] > 0
    assert "all-gather" in r["config"]["workload"] and r["n_gpus"] == 1


def test_rccl_ragged_gather_of_device_tensors():
    """vdetlib_amd.dist on the RCCL backend (world of one, forced): counts + padded payload of DEVICE tensors, an empty
    contribution, and gather_video_results == the local results."""
    import torch
    import torch.distributed as dist
    from vdetlib_amd import dist as vd
    assert not dist.is_initialized()
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "RANK")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", WORLD_SIZE="1", RANK="0")
    try:
        dev = torch.device("cuda", torch.cuda.current_device())
        vd.init(backend="nccl", device=dev, force=True)
        assert dist.get_backend() == "nccl"
        t = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3)
        parts = vd.all_gather_ragged(t, force=True)
        assert len(parts) == 1 and torch.equal(parts[0], t)
        empty = vd.all_gather_ragged(torch.zeros((0, 5), dtype=torch.int32, device=dev), force=True)
        assert len(empty) == 1 and tuple(empty[0].shape) == (0, 5)
        g = vd.all_gather_fixed(t, force=True)
        assert tuple(g.shape) == (1, 4, 3) and torch.equal(g[0], t)
        idx = torch.randint(0, 100, (3, 2, 4, 5), dtype=torch.int32, device=dev)
        cnt = torch.randint(0, 6, (3, 2, 4), dtype=torch.int32, device=dev)
        res = vd.gather_video_results([7, 2, 9], idx, cnt, force=True)
        assert sorted(res) == [2, 7, 9]
        for k, v in enumerate([7, 2, 9]):
            assert torch.equal(res[v][0], idx[k]) and torch.equal(res[v][1], cnt[k])
        torch.cuda.synchronize()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _gather_worker(rank, world, port, owners, serial, ret):
    """one gloo rank: contributes the results of the videos it owns, must end up with everybody's"""
    import torch
    import torch.distributed as dist
    from vdetlib_amd import dist as vd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    vd.init(backend="gloo")
    mine = owners[rank]
    shape_i, shape_c = next(iter(serial.values()))[0].shape, next(iter(serial.values()))[1].shape
    idx = torch.from_numpy(np.stack([serial[v][0] for v in mine])) if mine else torch.zeros((0,) + shape_i, dtype=torch.int32)
    cnt = torch.from_numpy(np.stack([serial[v][1] for v in mine])) if mine else torch.zeros((0,) + shape_c, dtype=torch.int32)
    res = vd.gather_video_results(mine, idx, cnt)
    ok = sorted(res) == sorted(v for o in owners for v in o)
    for v in res:
        ok = ok and np.array_equal(res[v][0].numpy(), serial[v][0]) and np.array_equal(res[v][1].numpy(), serial[v][1])
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_virtual_ranks_reproduce_the_serial_result():
    import torch
    from vdetlib_amd import ops, dist as vd
    C, K = 4, 640            # K >= the largest frame: every survivor list fits
    shapes = [(3 + (v % 4), 200 + 37 * v) for v in range(11)]          # (frames, boxes) per video
    vids = [synth.video(7000 + v, f, b, C) for v, (f, b) in enumerate(shapes)]

    def run(v):
        tb, ts = torch.from_numpy(vids[v][0]).cuda(), torch.from_numpy(vids[v][1]).cuda()
        idx, cnt = ops.nms_volume(tb, ts, 0.3, cap=K)
        Fmax = 6
        pi = torch.full((Fmax, C, K), -1, dtype=torch.int32, device='cuda'); pc = torch.zeros((Fmax, C), dtype=torch.int32, device='cuda')
        pi[:idx.shape[0]] = idx; pc[:cnt.shape[0]] = cnt
        return pi, pc

    serial = {v: run(v) for v in range(len(vids))}
    world = 8
    lpt_owners = vd.shard_lpt([f * b for f, b in shapes], world)
    for owners in ([vd.shard_round_robin(len(vids), r, world) for r in range(world)],
                   lpt_owners,
                   [[0, 3], [], [5, 1, 2], [], [], [4], [], []]):                   # ranks without a video
        if len(sum(owners, [])) == len(vids):
            assert sorted(v for o in owners for v in o) == list(range(len(vids)))   # a partition
        per_rank = []
        for mine in owners:
            res = [run(v) for v in mine]
            per_rank.append((mine, torch.stack([a for a, _ in res]) if res else torch.zeros((0, 6, C, K), dtype=torch.int32, device='cuda'),
                             torch.stack([b for _, b in res]) if res else torch.zeros((0, 6, C), dtype=torch.int32, device='cuda')))
        # the exchange itself: eight gloo ranks, each contributing what it owns (ragged, possibly nothing) through
        # dist.gather_video_results; every rank must end up with every video's serial result
        import socket
        import torch.multiprocessing as mp
        host = {v: (serial[v][0].cpu().numpy(), serial[v][1].cpu().numpy()) for v in serial}
        for k, (mine, gi, gc) in enumerate(per_rank):          # (what the rank computed on the GPU is what it will send)
            for j, v in enumerate(mine):
                assert torch.equal(gi[j], serial[v][0]) and torch.equal(gc[j], serial[v][1])
        if owners is lpt_owners:
            continue                                             # (two of the three shardings go through the process group)
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        ret = mp.Manager().dict()
        mp.spawn(_gather_worker, args=(world, port, [list(o) for o in owners], host, ret), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}
    lpt = vd.shard_lpt([f * b for f, b in shapes], world)
    loads = [sum(shapes[v][0] * shapes[v][1] for v in o) for o in lpt]
    rr = [sum(shapes[v][0] * shapes[v][1] for v in vd.shard_round_robin(len(vids), r, world)) for r in range(world)]
    assert max(loads) <= max(rr)


def _run_sharded(nproc, port, extra, env_extra=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "2",
           "--videos", "5", "--frames", "10", "--boxes", "1500", "--classes", "6", "--max-tracks", "3", "--streams", "2", "--cap", "512",
           "--no-cpu"] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_sharded_videos_world_of_one_rccl():
    """bench.py --videos (BASELINE configs[3] as written: LPT shard, videos in flight, ONE ragged exchange of all results per
    pass, one gathered video turned into protocol dicts on rank 0) over RCCL in a world of one rank"""
    r = _run_sharded(1, 29551, ["--force-exchange"])
    assert r["n_gpus"] == 1 and r["config"]["videos"] == 5 and r["config"]["shards"] == [[0, 1, 2, 3, 4]]
    x = r["exchange"]
    assert x["backend"] == "nccl" and x["world"] == 1 and x["all_videos_present"] is True and x["payload_bytes_per_rank"] > 0
    assert r["protocol_dicts"]["tracks"] > 0 and r["protocol_dicts"]["detections"] > 0
    assert abs(r["value"] - sum(r["config"]["frames"]) * 1500 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-6
    assert 0 < r["roofline"]["frac"] < 1 and len(r["per_rank"]["hbm_frac_algorithmic"]) == 1


def test_sharded_videos_two_ranks_dry_run():
    """the same with two ranks on one GPU (gloo): every video is processed by exactly one rank (LPT), all of them arrive"""
    r = _run_sharded(2, 29553, [], {"VDET_BENCH_ONE_GPU": "1"})
    sh = r["config"]["shards"]
    assert r["n_gpus"] == 2 and sorted(sh[0] + sh[1]) == [0, 1, 2, 3, 4] and sh[0] and sh[1]
    fr = r["config"]["frames"]
    assert abs(sum(fr[v] for v in sh[0]) - sum(fr[v] for v in sh[1])) <= max(fr)            # LPT balances the frames
    assert r["exchange"]["world"] == 2 and r["exchange"]["all_videos_present"] is True
    assert r["protocol_dicts"]["from_rank"] == 1 and r["protocol_dicts"]["tracks"] > 0
    assert len(r["per_rank"]["seconds"]) == 2 and r["scaling"] == "strong"


def test_sharded_16_videos_c2_quarter_oracle_checked():
    """configs[3] beyond the miniature: 16 videos of c2/4 size (56..94 frames x 10 000 boxes x 200 classes, 0.6 GB of scores each)
    through `bench.py --videos 16` in a world of one over RCCL -- LPT shard, 4 videos in flight, ONE ragged exchange per pass,
    one gathered video turned into protocol dicts and checked against the CPU oracle (kept lists of 4 frames x 200 classes,
    all tubelets of one class incl. re-scored boxes and pooled scores).  The default bench line runs the same at 64 videos of
    full c2 size (`sharded64`)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29557", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "2",
           "--videos", "16", "--frames", "75", "--no-cpu", "--force-exchange", "--check-oracle"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["config"]["videos"] == 16 and r["config"]["boxes"] == 10000 and r["config"]["classes"] == 200
    x = r["exchange"]
    assert x["backend"] == "nccl" and x["all_videos_present"] is True and x["videos_gathered"] == 16
    oc = r["oracle_check"]
    assert oc["nms_ok"] is True and oc["tubelets_ok"] is True and oc["nms_lists"] == len(oc["nms_frames"]) * 200 and oc["tubelets"] > 0
    assert len(r["lpt_loads_world8_boxes"]) == 8 and 1.0 <= r["lpt_imbalance_world8"] < 1.2
    assert r["protocol_dicts"]["tracks"] > 0 and r["protocol_dicts"]["detections"] > 0


def test_bench_eight_ranks_one_gpu_dry_run():
    """the driver's 8-GPU launch line -- torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 -- with all eight ranks on
    device 0 over gloo (VDET_BENCH_ONE_GPU=1, miniature videos): every rank steps its own videos, the exchange gathers eight
    slots, rank 0 prints the one line with eight per-rank times; and the same for `--videos 19` (LPT over eight ranks)"""
    env = dict(os.environ, VDET_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1"]
    cmd = base + ["--master-port", "29561", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "2",
                  "--frames", "8", "--boxes", "1200", "--classes", "8", "--streams", "2", "--cap", "512", "--no-cpu"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["scaling"] == "weak" and len(r["per_rank"]["ms_per_step"]) == 8
    assert abs(r["value"] - 8 * 8 * 1200 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-6
    assert r["exchange"]["world"] == 8 and r["exchange"]["own_slot_matches"] is True
    cmd = base + ["--master-port", "29563", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--videos", "19",
                  "--frames", "8", "--boxes", "1200", "--classes", "6", "--max-tracks", "3", "--streams", "2", "--cap", "512", "--no-cpu"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    sh = r["config"]["shards"]
    assert r["n_gpus"] == 8 and len(sh) == 8 and sorted(v for o in sh for v in o) == list(range(19)) and all(sh)
    assert r["exchange"]["world"] == 8 and r["exchange"]["all_videos_present"] is True and r["exchange"]["videos_gathered"] == 19
    assert len(r["per_rank"]["seconds"]) == 8 and r["protocol_dicts"]["from_rank"] == 7
