"""-m gpu: BASELINE configs[3] (videos sharded across GPUs + one exchange step) as far as ONE GPU can take it:
(a) bench.py's N = 2 control flow end to end -- two ranks launched by torch.distributed.run, both on device 0,
    exchange over gloo (VDET_BENCH_ONE_GPU=1): barriers, the all-gather inside the timed step, MAX over ranks;
(b) eight "virtual ranks" in one process: shard_round_robin / shard_lpt + per-video results + the gather
    reproduce the serial result for 11 videos of different sizes.
The RCCL transport itself (one rank per GPU over xGMI) needs the driver's multi-GPU node."""
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
    for owners in ([vd.shard_round_robin(len(vids), r, world) for r in range(world)],
                   vd.shard_lpt([f * b for f, b in shapes], world)):
        assert sorted(v for o in owners for v in o) == list(range(len(vids)))       # a partition
        per_rank = []
        for mine in owners:
            res = [run(v) for v in mine]
            per_rank.append((mine, torch.stack([a for a, _ in res]) if res else torch.zeros((0, 6, C, K), dtype=torch.int32, device='cuda'),
                             torch.stack([b for _, b in res]) if res else torch.zeros((0, 6, C), dtype=torch.int32, device='cuda')))
        # what all_gather_ragged delivers on every rank = the per-rank tensors in rank order
        merged = {}
        for mine, gi, gc in per_rank:
            for k, v in enumerate(mine):
                merged[v] = (gi[k], gc[k])
        assert sorted(merged) == sorted(serial)
        for v in serial:
            assert torch.equal(merged[v][0], serial[v][0]) and torch.equal(merged[v][1], serial[v][1])
    lpt = vd.shard_lpt([f * b for f, b in shapes], world)
    loads = [sum(shapes[v][0] * shapes[v][1] for v in o) for o in lpt]
    rr = [sum(shapes[v][0] * shapes[v][1] for v in vd.shard_round_robin(len(vids), r, world)) for r in range(world)]
    assert max(loads) <= max(rr)
