"""world_size-2 gloo tests of the data-parallel plumbing (pairs sharded round-robin, no data-path collective,
MAX-reduced timing and SUM-reduced PSNR)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "fldr-vfi_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fldr_harness as Hn
    mine = Hn.shard_pairs(7, rank, world)
    # each rank "processes" its pairs: per-pair PSNR = 30 + pair index, step time = 0.1 * (rank + 1)
    psum, cnt = sum(30.0 + i for i in mine), len(mine)
    mean, n = Hn.reduce_psnr(psum, cnt)
    tmax = Hn.max_over_ranks(0.1 * (rank + 1))
    dist.barrier()
    q.put((rank, mine, mean, n, tmax))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reductions():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, mean0, n0, t0), (r1, m1, mean1, n1, t1) = res
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]                 # round-robin over PAIRS, disjoint, complete
    assert n0 == n1 == 7
    assert mean0 == pytest.approx(33.0) and mean1 == pytest.approx(33.0)
    assert t0 == pytest.approx(0.2) and t1 == pytest.approx(0.2)   # max over ranks


def test_single_process_helpers_are_noops():
    import fldr_harness as Hn
    assert Hn.shard_pairs(5, 0, 1) == [0, 1, 2, 3, 4]
    assert Hn.max_over_ranks(1.5) == 1.5
    assert Hn.reduce_psnr(90.0, 3) == (30.0, 3)
    assert 1 <= Hn.host_cores() <= 64
