"""CPU-only: the N > 1 path (video sharding + the single result exchange) with world_size 2 on the
gloo backend -- same code that runs on RCCL across 8 MI355X."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vdetlib_amd import dist as vd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_result(video_id, F, C, K):
    rng = np.random.RandomState(video_id)
    cnt = rng.randint(0, K + 1, (F, C)).astype(np.int32)
    idx = rng.randint(0, 1000, (F, C, K)).astype(np.int32)
    return torch.from_numpy(idx), torch.from_numpy(cnt)


def _worker(rank, world, port, n_videos, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    w, r, _ = vd.init(backend="gloo")
    assert (w, r) == (world, rank)
    mine = vd.shard_round_robin(n_videos, rank, world)
    F, C, K = 3, 4, 5
    res = [_fake_result(v, F, C, K) for v in mine]
    idx = torch.stack([a for a, _ in res]) if res else torch.zeros((0, F, C, K), dtype=torch.int32)
    cnt = torch.stack([b for _, b in res]) if res else torch.zeros((0, F, C), dtype=torch.int32)
    allres = vd.gather_video_results(mine, idx, cnt)
    ok = sorted(allres) == list(range(n_videos))
    for v in range(n_videos):
        a, b = _fake_result(v, F, C, K)
        ok = ok and torch.equal(allres[v][0], a) and torch.equal(allres[v][1], b)
    # ragged gather with an empty rank
    t = torch.arange(rank * 3, dtype=torch.float32).reshape(-1, 1)
    parts = vd.all_gather_ragged(t)
    ok = ok and [p.shape[0] for p in parts] == [r_ * 3 for r_ in range(world)]
    # fixed-shape gather (what bench.py --gpus N uses: one collective, no count exchange / host sync)
    fx = torch.full((2, 3), float(rank), dtype=torch.float32)
    g = vd.all_gather_fixed(fx)
    ok = ok and tuple(g.shape) == (world, 2, 3) and all(bool((g[r_] == float(r_)).all()) for r_ in range(world))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_videos", [5, 1])
def test_world2_gloo_gather(n_videos):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_videos, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_class_slices_partition():
    for C, world in ((200, 8), (30, 8), (5, 8), (201, 4)):
        got = [vd.class_slice(C, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == C
        assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in got) - min(b - a for a, b in got) <= 1


def test_sharding_rules():
    assert vd.shard_round_robin(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((vd.shard_round_robin(64, r, 8) for r in range(8)), [])) == list(range(64))
    owned = vd.shard_lpt([300 * 10000, 30 * 300, 2000 * 300, 300 * 10000, 50 * 300, 700 * 2000], 2)
    assert sorted(owned[0] + owned[1]) == list(range(6))
    loads = [sum([300 * 10000, 30 * 300, 2000 * 300, 300 * 10000, 50 * 300, 700 * 2000][i] for i in o) for o in owned]
    assert abs(loads[0] - loads[1]) <= 700 * 2000
    assert vd.all_gather_ragged(torch.ones(3, 2))[0].shape == (3, 2)      # single process: identity


def test_forced_exchange_in_a_world_of_one_gloo():
    """dist.init(force=True): a single process still gets its process group, and the forced collectives run (the path
    bench.py --force-exchange takes with RCCL on a single-GPU box)."""
    assert not dist.is_initialized()
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "RANK")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1", RANK="0")
    try:
        w, r, _ = vd.init(backend="gloo", force=True)
        assert (w, r) == (1, 0) and dist.is_initialized() and dist.get_world_size() == 1
        t = torch.arange(6, dtype=torch.float32).reshape(3, 2)
        assert vd.all_gather_fixed(t)[0].data_ptr() == t.data_ptr()              # not forced: the identity, no collective
        g = vd.all_gather_fixed(t, force=True)
        assert tuple(g.shape) == (1, 3, 2) and torch.equal(g[0], t) and g.data_ptr() != t.data_ptr()
        parts = vd.all_gather_ragged(torch.zeros((0, 4), dtype=torch.int32), force=True)
        assert len(parts) == 1 and tuple(parts[0].shape) == (0, 4)
        idx = torch.randint(0, 9, (2, 3, 4, 5), dtype=torch.int32)
        cnt = torch.randint(0, 6, (2, 3, 4), dtype=torch.int32)
        res = vd.gather_video_results([4, 1], idx, cnt, force=True)
        assert sorted(res) == [1, 4] and torch.equal(res[4][0], idx[0]) and torch.equal(res[1][1], cnt[1])
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_frame_offsets_are_validated():
    from vdetlib_amd import ops
    assert ops._frame_offsets([0, 3, 7], 7).tolist() == [0, 3, 7]
    for bad in ([1, 3, 7], [0, 3, 3, 7], [0, 8], [0], [0, 3, 6]):
        with pytest.raises(ValueError):
            ops._frame_offsets(bad, 7)


def test_bench_gpus_n_without_devices_exits_2():
    """`python bench.py --gpus N` launches N ranks by itself (bench.self_launch); with fewer than N devices visible -- this
    container has none -- it exits with code 2 and says why, instead of printing an n_gpus: 1 line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VDET_BENCH_ONE_GPU")}
    import torch
    if torch.cuda.device_count() >= 9:
        pytest.skip("more devices than any node has")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "9", "--steps", "1", "--warmup", "1"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "--gpus 9" in p.stderr and "visible" in p.stderr
    assert "{" not in p.stdout


def _packed_worker(rank, world, port, ret):
    """PackedExchange (the configs[3] exchange: rounds of one fixed-capacity all-gather) on two gloo ranks: rank 0 owns videos
    {0, 2, 4} of 3 / 5 / 2 frames, rank 1 owns {1, 3} -- so round 2 carries an EMPTY record from rank 1 -- and two passes reuse the
    buffers."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    vd.init(backend="gloo")
    frames = {0: 3, 1: 4, 2: 5, 3: 1, 4: 2}
    owned = [[0, 2, 4], [1, 3]]
    rounds = max(len(o) for o in owned)

    def fields(v, salt):
        f = frames[v]
        return [torch.arange(f * 7, dtype=torch.float32) + 100 * v + salt, torch.full((f, 2), v + salt, dtype=torch.int32)]
    cap = 5 * 7 * 4 + 16 + 5 * 2 * 4 + 16
    x = vd.PackedExchange(cap, rounds, "cpu")
    mine = owned[rank]
    for j in range(rounds):
        if j < len(mine):
            x.set_record(j, mine[j], frames[mine[j]], [t.numel() * 4 for t in fields(mine[j], 0)])
        else:
            x.set_record(j, -1, 0, [])
    ok = True
    for salt in (0, 1000):                       # two passes through the same buffers
        for j in range(rounds):
            if j < len(mine):
                x.pack(j, fields(mine[j], salt))
            x.launch(j)
        x.join()
        for j in range(rounds):
            for r_ in range(world):
                hdr, fl = x.record_of(j, r_)
                if j < len(owned[r_]):
                    v = owned[r_][j]
                    want = fields(v, salt)
                    ok = ok and hdr[:3] == [v, frames[v], 2]
                    ok = ok and torch.equal(fl[0].view(torch.float32), want[0]) and torch.equal(fl[1].view(torch.int32).reshape(-1, 2), want[1])
                else:
                    ok = ok and hdr[0] == -1 and fl == []
    try:
        x.set_record(0, 9, 99, [cap + 64])
        ok = False
    except ValueError:
        pass
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_packed_exchange_rounds():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_packed_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}
