"""CPU, world_size 2, gloo: the N > 1 result path of bench.py (gather of variable-length match lists to rank 0)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mono_slam_framework_amd.gather import MatchListGather, shard_pairs


def _fake_lists(rank, P):
    rng = np.random.default_rng(100 + rank)
    counts = rng.integers(0, 7, size=P)
    if rank == 1:
        counts[:] = 0 if P < 3 else counts     # exercise an empty rank when P is small
    offs = np.zeros(P + 1, np.int32)
    offs[1:] = np.cumsum(counts)
    packed = rng.integers(0, 1280, size=(int(offs[-1]) + 3, 4)).astype(np.int32)   # tail garbage must not travel
    return packed, offs


def _worker(rank, world, port, P, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = MatchListGather(P, torch.device("cpu"))
        for step in range(2):
            packed, offs = _fake_lists(rank, P)
            res = g(torch.from_numpy(packed), torch.from_numpy(offs))
            if rank == 0:
                ok = len(res) == world
                for r in range(world):
                    ep, eo = _fake_lists(r, P)
                    ok = ok and np.array_equal(res[r][1].numpy(), eo) and np.array_equal(res[r][0].numpy(), ep[:eo[-1]])
                q.put(ok)
            else:
                assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(P):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, P, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) and q.get(timeout=5)


def test_gather_two_ranks():
    _run(16)


def test_gather_with_empty_rank():
    _run(2)


def test_shard_is_a_partition():
    parts = [shard_pairs(37, r, 4) for r in range(4)]
    assert sum(parts, []) == list(range(37))                      # contiguous blocks in rank order
    assert [len(p) for p in parts] == [10, 10, 10, 7]
    assert shard_pairs(3, 3, 4) == [] and shard_pairs(3, 0, 4) == [0]
