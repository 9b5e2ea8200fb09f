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


def _run_bench(extra, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, ([json.loads(l) for l in lines])


def test_bench_self_launches_n_ranks_gloo_dry():
    """`python bench.py --gpus 2` with no torch.distributed environment starts the two ranks itself (the parent never
    touches the GPU); --backend gloo --dry runs the same launcher / barrier / MAX-reduction / JSON path on CPU."""
    r, js = _run_bench(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(js) == 1, r.stdout                          # rank 0 prints ONE line
    j = js[0]
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1 and j["scaling"] == "weak" and j["dry"] is True
    assert len(j["config"]["per_rank_pairs_per_s"]) == 2
    assert j["value"] == pytest.approx(2 * 4 / (j["ms_per_step"] * 4e-3), rel=1e-3)
    assert j["value"] < sum(j["config"]["per_rank_pairs_per_s"]) * 1.001        # whole-job rate uses the MAX over ranks
    # the CPU-baseline leg runs on rank 0 at ANY world size (after the other ranks were released), so a multi-GPU line is complete
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["runs"] == 1 and cb["value"] > 0 and cb["cores"] >= 1


def test_bench_launcher_failure_modes():
    r, js = _run_bench(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "0"],
                       {"FLDR_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0 and not js                     # a failing child fails the parent, no result line
    r, js = _run_bench(["--gpus", "2", "--dry"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_bench_forced_process_group_world_size_one_gloo_dry():
    """FLDR_BENCH_FORCE_PG=1 under `torch.distributed.run --nproc-per-node 1`: the process group, barriers, MAX reduction and
    all_gather of the multi-GPU path run at world size 1 (CPU twin — gloo, --dry — of the RCCL test in tests/test_gpu_parity.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FLDR_BENCH_FORCE_PG"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--backend", "gloo", "--dry", "--steps", "3",
           "--warmup", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    js = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(js) == 1 and js[0]["n_gpus"] == 1
    pg = js[0]["config"]["process_group"]
    assert pg["backend"] == "gloo" and pg["world_size"] == 1 and pg["forced_at_world_size_1"] is True
    assert len(js[0]["config"]["per_rank_pairs_per_s"]) == 1
