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


def test_bench_self_launches_eight_ranks_gloo_dry():
    """The N = 8 launch the driver makes on an 8-GPU node, rehearsed on CPU (gloo, --dry: launcher, rendezvous on 127.0.0.1, barriers,
    MAX-reduction, all_gather of the per-rank rates, rank 0's complete JSON line with its CPU-baseline leg): eight ranks, eight disjoint
    shards, one line.  No scaling curve has been measured on hardware."""
    import fldr_harness as Hn
    r, js = _run_bench(["--gpus", "8", "--backend", "gloo", "--dry", "--steps", "4", "--warmup", "1"], timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(js) == 1, r.stdout
    j = js[0]
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["dry"] is True
    assert len(j["config"]["per_rank_pairs_per_s"]) == 8 and all(x > 0 for x in j["config"]["per_rank_pairs_per_s"])
    assert j["config"]["parallelism"].startswith("dp8")
    assert j["value"] == pytest.approx(8 * 4 / (j["ms_per_step"] * 4e-3), rel=1e-3)
    assert 1 <= j["config"]["setup_threads_per_rank"] <= max(1, Hn.host_cores() // 8) + 1      # each rank takes its share of the host
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["runs"] == 1 and cb["value"] > 0
    # the shards the eight ranks draw their pairs from: disjoint and complete
    shards = [Hn.shard_pairs(8 * 4, rk, 8) for rk in range(8)]
    assert sorted(sum(shards, [])) == list(range(32)) and all(len(s) == 4 for s in shards)


def test_bench_eight_ranks_one_killed_fails_the_parent_within_the_deadline():
    import time
    t0 = time.time()
    r, js = _run_bench(["--gpus", "8", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "0"], {"FLDR_BENCH_TEST_FAIL_RANK": "5"}, timeout=300)
    assert r.returncode != 0 and not js
    assert time.time() - t0 < 240


def _eval_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "fldr-vfi_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fldr_harness as Hn
    n_pairs = 13                                                   # uneven: ranks 0-4 get two pairs, ranks 5-7 one
    mine = Hn.shard_pairs(n_pairs, rank, world)
    # the reduction evaluate_dir performs at its end (SUM of (sum PSNR, sum SSIM, count) over the ranks), on synthetic per-pair scores:
    # 7 outputs per pair, PSNR = 30 + pair, SSIM = 0.9 + pair / 1000
    psum = sum(7 * (30.0 + i) for i in mine)
    ssum = sum(7 * (0.9 + i / 1000.0) for i in mine)
    cnt = 7 * len(mine)
    mean_p, n = Hn.reduce_psnr(psum, cnt)
    mean_s, n2 = Hn.reduce_psnr(ssum, cnt)
    tmax = Hn.max_over_ranks(0.01 * (rank + 1))
    rates = Hn.gather_floats(100.0 + rank)
    dist.barrier()
    q.put((rank, mine, mean_p, mean_s, n, n2, tmax, rates))
    dist.destroy_process_group()


def test_eight_rank_evaluation_reduction_uneven_pairs():
    """The reductions of fldr_harness.evaluate_dir (main.py:885-911's averages over the data set, computed shard by shard) at world
    size 8 with a pair count that does not divide: every rank ends with the same global means and count."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_pairs = sorted(sum((r[1] for r in res), []))
    assert all_pairs == list(range(13))
    assert [len(r[1]) for r in res] == [2, 2, 2, 2, 2, 1, 1, 1]
    want_p = sum(30.0 + i for i in range(13)) / 13
    want_s = sum(0.9 + i / 1000.0 for i in range(13)) / 13
    for r in res:
        assert r[2] == pytest.approx(want_p) and r[3] == pytest.approx(want_s) and r[4] == r[5] == 7 * 13
        assert r[6] == pytest.approx(0.08)
        assert [round(x) for x in r[7]] == [100 + k for k in range(8)]
