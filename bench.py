#!/usr/bin/env python3
"""Headline benchmark: 4K (3840x2160) frame pairs interpolated per second (BASELINE.json metric, config 2:
single 4K frame pair, 5-scale test path, t = 0.5) through the MI355X-native hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one (frame pair, t) forward = the reference's `model_net(input_gpuList, t, normInput=pyramid, ...)` call
(main.py:867) with the 6-level pyramid already resident in HBM.  Frame pairs are independent, so N ranks each
interpolate their own pairs (weak scaling, no data-path collective); the only RCCL traffic is the barrier, a
max-reduction of the elapsed time and a gather of the per-rank rates.  Rank 0 prints ONE JSON line.

`--gpus N` without a torch.distributed environment launches the N ranks itself (torch.distributed.run as a CHILD
process, before this process touches the GPU) and exits with the child's code; inside a launch whose WORLD_SIZE
differs from --gpus it refuses to run.  `--backend gloo --dry` exercises launcher, barriers, reductions and the
JSON line on CPU only (tests/test_distributed_cpu.py).

The timed loop rotates DISTINCT frame pairs (>= 4, one per in-flight stream and more): consecutive steps never read
the same pyramid, and the pyramids together (4 x 377 MB) exceed the 256 MiB Infinity Cache.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fldr-vfi_amd"))

H4K, W4K = 2160, 3840
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_FP16_MFMA_TFLOPS = 2500.0      # dense fp16/bf16 matrix peak
PEAK_HBM_GBPS = 8000.0
# SURVEY.md App. C / BASELINE.md section 3: compulsory HBM traffic and algorithmic FLOPs of ONE forward
PATH_MODEL = {(2160, 3840): {"hbm_bytes": 5.284e9, "flops": 0.314e12}, (2160, 4096): {"hbm_bytes": 5.636e9, "flops": 0.335e12}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent frame-pair forwards kept in flight on separate HIP streams (1 = strictly serial)")
    ap.add_argument("--pairs", type=int, default=4, help="distinct frame pairs rotated through the timed loop (>= streams)")
    ap.add_argument("--sustained-s", type=float, default=3.0, help="length of the extra sustained-rate measurement (0 = skip)")
    ap.add_argument("--varying-motion-steps", type=int, default=60, help="steps of the informational run on pairs with a smoothly varying motion field (0 = skip)")
    ap.add_argument("--incl-ingest-steps", type=int, default=60, help="steps of the informational uint8-in -> uint8-out run with the ingest kernels inside the loop (0 = skip)")
    ap.add_argument("--multi-t-pairs", type=int, default=4, help="pairs of the informational 4096x2160 7-outputs-per-pair run, BASELINE config 3 (0 = skip)")
    ap.add_argument("--fp16-mode-steps", type=int, default=60, help="steps of the informational fp16-input convolution run (BASELINE config 5; 0 = skip)")
    ap.add_argument("--config5-steps", type=int, default=45, help="steps of the informational BASELINE config 5 run (4096x2160, fp16-input convolutions, with its own roofline; 0 = skip)")
    ap.add_argument("--xtest-dir", default=None, help="X-Test style folder (<dir>/<type>/<scene>/*.png, 33 frames per scene): after the timed region every rank "
                                                      "evaluates its share of the pairs (8x: 7 outputs per pair) and parity.x_test_psnr / x_test carry the mean PSNR / SSIM-Y")
    ap.add_argument("--xtest-multiple", type=int, default=8)
    ap.add_argument("--no-graphs", action="store_true", help="enqueue every step eagerly (default: each (stream, pair) forward is captured once in a hipGraph and replayed: "
                                                             "the same kernels, ~0.03 instead of ~1 ms of host time per step)")
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry", action="store_true", help="no GPU work: a step is a 1 ms sleep (CPU test of the launcher / reductions)")
    ap.add_argument("--share-gpu", action="store_true", help="REHEARSAL on a 1-GPU box: all N ranks run their GPU work on device 0 (process group: gloo); exercises the "
                                                             "multi-process launch, per-rank sharding, barriers and reductions around real HIP work - the line says so and its value is no measurement")
    return ap.parse_args()


def self_launch(a):
    """Spawn the N ranks as a child `python -m torch.distributed.run`.  Nothing in this process has touched the GPU
    (torch is not even imported yet); the parent only waits and forwards the exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FLDR_BENCH_CHILD"] = "1"
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------
def dominant_conv_roofline(model, h, w, device, steps, precision="split"):
    """Roofline of the dominant kernel: the 96->96 3x3 convolution at the level-0 feature map (rec_ctx_ds.0/.2,
    conv_flow2.0/.2 launch this instance of conv3x3_ring_kernel<3,3,false,8>; the 3x3 convolutions are ~30 % of the GPU time).
    algorithmic FLOPs per launch = 2 * cin * cout * 9 * pixels (165 888 MAC/px for the two
    rec_ctx_ds convolutions); `achieved` = algorithmic FLOPs / average launch time, `frac` = that over the dense
    fp16 MFMA peak (= frac_algorithmic).  The kernel forms every product from THREE fp16 MFMAs (hi*hi + hi*lo + lo*hi,
    fp32-equivalent accuracy), so the matrix pipe executes 3x the algorithmic flops: `frac_issued`.  Duration: HIP
    events on the launch stream, operands split-packed in HBM as inside the model.  The exact fp32-MFMA kernel of the
    same layer (FLDR_CONV_PRECISION=fp32) is timed beside it."""
    import torch
    import fldr_hip
    x = torch.rand(1, 96, h, w, device=device) * 2 - 1
    xp = fldr_hip.spk_pack(x)
    conv = model.rec_ctx_ds[0]
    out = torch.empty(1, 96, h, w, device=device)
    n = max(20, steps)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    ms = timed(lambda: fldr_hip.conv2d_spk([xp], conv.weight, conv.bias, relu=True, want_f32=False, want_spk=True, precision=precision))
    flops = 2.0 * 96 * 96 * 9 * h * w
    alg = flops / (ms * 1e-3) / 1e12
    if precision == "fp16":
        # BASELINE config 5: ONE fp16 MFMA per product (hi halves only: the lo planes are neither loaded nor multiplied) — issued == algorithmic
        return {"bound": "mfma", "kernel": "conv3x3_ring_kernel<3,1,false,8> (3x3 96->96 @%dx%d, the same ring on the hi halves only: plain fp16 inputs, "
                                           "one v_mfma_f32_16x16x32_f16 per product, fp32 accumulation)" % (h, w),
                "achieved": round(alg, 1), "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(alg / PEAK_FP16_MFMA_TFLOPS, 4),
                "frac_issued": round(alg / PEAK_FP16_MFMA_TFLOPS, 4), "traffic": None, "launch_ms": round(ms, 4), "flops_per_launch": flops,
                "algorithmic_bytes_per_launch": 2 * 96 * h * w * 4,
                "achieved_GBps_on_algorithmic_bytes": round(2 * 96 * h * w * 4 / (ms * 1e-3) / 1e9, 1)}
    ms32 = timed(lambda: fldr_hip.conv2d([x], conv.weight, conv.bias, relu=True, out=out, precision="fp32"))
    tr, tr_src = _measured_traffic()
    return {"bound": "mfma", "kernel": "conv3x3_ring_kernel<3,3,false,8> (3x3 96->96 @%dx%d, persistent loader/consumer ring, split-packed operands, "
                                       "3 x fp16-split v_mfma_f32_16x16x32_f16)" % (h, w),
            "achieved": round(alg, 1), "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(alg / PEAK_FP16_MFMA_TFLOPS, 4),
            "frac_algorithmic": round(alg / PEAK_FP16_MFMA_TFLOPS, 4), "frac_issued": round(3 * alg / PEAK_FP16_MFMA_TFLOPS, 4),
            "frac_of_fp32_mfma_peak": round(alg / PEAK_FP32_MFMA_TFLOPS, 3),
            "traffic": tr, "traffic_source": tr_src, "launch_ms": round(ms, 4), "flops_per_launch": flops, "issued_mfma_flops_per_launch": 3.0 * flops,
            "algorithmic_bytes_per_launch": 2 * 96 * h * w * 4,
            "exact_fp32_mfma_kernel": {"launch_ms": round(ms32, 4), "achieved": round(flops / (ms32 * 1e-3) / 1e12, 1),
                                       "peak": PEAK_FP32_MFMA_TFLOPS, "frac": round(flops / (ms32 * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}}


def _moved_bytes_per_forward():
    """HBM-side bytes a warm 4K forward actually moves (sum of FETCH_SIZE x 2 + WRITE_SIZE over its launches) from the newest
    committed forward profile (tools/prof_forward_pmc.sh + pmc_forward_summary.py); older summaries carry no total: summed here
    without their one-time prepack / harness kernels."""
    import re
    skip = re.compile(r"prepack|absmax|at::native|__amd_rocclr|FillFunctor|direct_copy|elementwise_kernel")
    for name in ("r06_forward_pmc.json", "r05_forward_pmc.json", "r04_forward_pmc.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            mb = d.get("moved_MB_per_forward")
            if mb is None:
                mb = sum(k["fetch_x2_MB"] + k["write_MB"] for k in d["kernels"] if not skip.search(k["kernel"]))
            return mb * 1e6, "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over single-stream 3840x2160 forwards; not measured in this run)" % name
        except Exception:
            pass
    return None, None


def _measured_traffic():
    """HBM-side bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; separate --pmc passes); newest round first."""
    for name in ("r06_conv96_spk_traffic.json", "r05_conv96_spk_traffic.json", "r04_conv96_spk_traffic.json", "r03_conv96_spk_traffic.json", "r02_conv96_spk_traffic.json", "r01_conv96_spk_traffic.json"):
        try:
            v = json.load(open(os.path.join(ROOT, "profiles", name)))["hbm_bytes_per_launch"]
            return v, "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_round.sh; not measured in this run)" % name
        except Exception:
            pass
    return None, None


def _energy_record():
    """Joules per pair of the sustained loop and board power from the newest committed energy table (tools/energy_table.sh ->
    profiles/r06_energy_by_kernel.json: every stage of the forward looped while rocm-smi is sampled; the loop's power against the
    board's cap is the bound that binds this path)."""
    for name in ("r06_energy_by_kernel.json",):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            sm = d["summary"]
            top = sorted((r for r in d["rows"] if r["stage"] not in ("idle",) and not r["stage"].startswith(("whole forward", "sustained"))),
                         key=lambda r: -r["mJ_dynamic"])[:3]
            return {"joules_per_pair": sm["joules_per_pair"], "watts_sustained": sm["watts_sustained"], "cap_watts": d.get("cap_watts", 1400.0),
                    "idle_watts": sm["idle_watts"], "ms_per_step_when_measured": sm["ms_per_step"], "shader_mhz_sustained": sm["mhz_sustained"],
                    "stage_model_over_measured": sm["model_over_measured"],
                    "top_dynamic_mJ": {r["stage"]: round(r["mJ_dynamic"], 1) for r in top},
                    "source": "profiles/%s (tools/energy_table.sh; not measured in this run)" % name}
        except Exception:
            pass
    return None


def host_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    from fldr_harness import host_core_budget
    return dict(host_core_budget(), cpu_model=model)


def cpu_baseline(frames_cpu, t_cpu, gpu_out=None, runs=3):
    """The oracle (CPU restatement of the reference, kind 'port') timed on this box's host cores on the SAME 4K pair:
    1 warm-up + `runs` timed forwards with every core this process may use (3 at one GPU; 1 on rank 0 of a multi-GPU
    launch, whose other ranks have left by then), then 1 + 2 with 8 threads (the survey container's count; BASELINE.md
    section 4; one-GPU runs only); pyramid excluded like the GPU number.  The warm-up's output doubles as the full-size
    parity check of the GPU frame (PSNR of the rounded 8-bit frames, max abs error)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fldr_oracle as O
    import fldr_harness as Hn
    info = host_info()
    cores = info["cores_used"]
    w = O.load_weights(Hn.DEFAULT_WEIGHTS)
    pyr = O.pad_and_pyramid(frames_cpu)
    H, W = frames_cpu.shape[-2:]

    def run(nthreads, nruns):
        torch.set_num_threads(nthreads)
        ts, ref = [], None
        with torch.no_grad():
            for i in range(nruns + 1):
                t0 = time.time()
                ref = O.forward(w, pyr, t_cpu)
                ts.append(time.time() - t0)
        return ts[1:], ref

    ts, ref = run(cores, runs)
    mean = sum(ts) / len(ts)
    res = {"value": round(1.0 / mean, 5), "unit": "4K frame-pairs/s", "cores": cores, "kind": "port", "runs": runs,
           "sample": "oracle/fldr_oracle.py (torch-CPU) on the same %dx%d pair (rank 0's first pair, t=0.5): 1 warm-up + %d timed forward(s), %s s"
                     % (W, H, runs, " / ".join("%.2f" % x for x in ts)),
           "host": info}
    if cores != 8 and runs >= 3:
        ts8, _ = run(min(8, cores), 2)
        res["at_8_threads"] = {"value": round(len(ts8) / sum(ts8), 5), "runs_s": [round(x, 2) for x in ts8]}
    parity = None
    if gpu_out is not None:
        ref = ref[:, :, :H, :W]
        err = (gpu_out.double() - ref.double()).abs()
        parity = {"vs": "oracle (CPU restatement pinned bit-for-bit by the reference's golden frames)", "max_abs_err": float(err.max()),
                  "mean_abs_err": float(err.mean()),
                  "psnr_8bit_db": Hn.psnr(Hn.to_uint8_image(ref[0]), Hn.to_uint8_image(gpu_out[0])),
                  "x_test_psnr": None,
                  "note": "X-Test / Xiph / Inter4K are not available offline: PSNR on X-Test cannot be reported; the rounded 8-bit "
                          "frame of this synthetic 4K pair is compared with the oracle's instead"}
    return res, parity


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(self_launch(a))                         # parent: no GPU call, no torch import
    world = int(env_world or "1")
    if world != a.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: launch with matching values (or without a torch.distributed "
              "environment: bench.py starts the ranks itself)" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FLDR_BENCH_TEST_FAIL_RANK") == str(rank):     # launcher test: a failing rank must fail the parent
        sys.exit(3)

    import torch
    import torch.distributed as dist
    from fldr_harness import shard_pairs, max_over_ranks, gather_floats, host_core_budget
    # N ranks share the box's cores while they build their synthetic 4K pairs (torch-CPU): each takes its share, not all of them
    setup_threads = max(1, host_core_budget()["cores_used"] // max(world, 1))
    torch.set_num_threads(setup_threads)
    # FLDR_BENCH_FORCE_PG=1: create the process group (and run every barrier / reduction / gather of the multi-GPU path) at world
    # size 1 too — the RCCL rehearsal a 1-GPU box allows (tests/test_gpu_parity.py::test_bench_rccl_process_group_world_size_one)
    use_pg = world > 1 or os.environ.get("FLDR_BENCH_FORCE_PG") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    gpu = not a.dry
    if gpu:
        dev_index = 0 if a.share_gpu else local_rank
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    else:
        device = torch.device("cpu")
    pg_backend = None
    if use_pg:
        pg_backend = "nccl" if (a.backend == "nccl" and gpu and not a.share_gpu) else "gloo"       # (RCCL refuses two ranks on one device)
        if pg_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)              # RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    def sync():
        if gpu:
            torch.cuda.synchronize()

    def barrier():
        if use_pg:
            dist.barrier()

    npairs = max(a.pairs, a.streams, 1)
    my_pairs = shard_pairs(world * npairs, rank, world)          # pair index = seed; disjoint across ranks
    latency_ms = latency_eager_ms = dt_e2e = sustained = fp16_mode = varying = incl = multi_t = config5 = None
    if gpu:
        import fldr_harness as Hn
        model, _, args = Hn.prepare_model(device)
        t = torch.tensor([[0.5]], device=device)
        frames, pyrs = [], []
        with torch.no_grad():
            for p in my_pairs:
                f = Hn.frames_from_uint8(Hn.synthetic_pair(a.height, a.width, seed=p)).to(device)
                frames.append(f)
                pyrs.append(Hn.build_pyramid(Hn.pad_frames(f, args), args))
        pyr_bytes = sum(x.numel() * 4 for x in pyrs[0])
        # Frame pairs are independent, so a serving loop keeps several in flight: step i is enqueued on HIP stream
        # i % streams and interpolates pair i % npairs.  Every step is a complete forward; overlapping them fills the
        # CUs that the latency-bound coarse pyramid levels (9x15 ... 72x120 feature maps) leave idle.
        streams = [torch.cuda.Stream(device=device) for _ in range(max(1, a.streams))]
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream())

        def eager_step(i):
            k = i % npairs
            with torch.cuda.stream(streams[i % len(streams)]), torch.no_grad():
                return Hn.interpolate(model, args, frames[k], t, pyramid=pyrs[k])

        # hipGraph replay (the contract's "capture launch-bound inner loops in hipGraphs"): after the eager priming below, the forward
        # of every (stream, pair) combination is captured ONCE (the same kernels through the same C ABI on that stream; one memory pool
        # per stream) and a step is one replay: ~0.03 ms of host time instead of ~1 ms (64 launches from Python), which keeps the loop
        # GPU-bound on hosts that are slow or busy.  On a fast host both forms give the same rate (tools/graph_probe.py: 2.03 vs 2.04 ms;
        # replayed frames == eager frames bit for bit).  Any capture failure falls back to eager steps and is reported.
        graphs, graph_outs, graph_state = {}, {}, {"on": False, "why": "disabled (--no-graphs)" if a.no_graphs else "not captured yet"}

        def capture_graphs():
            # fldr_harness.GraphedInterpolator (the opt-in replay API of the harness): one instance per (stream, pair), the instances of
            # a stream share a memory pool; the first replay of each is checked against an eager forward of its pair on its stream (the same
            # bits, or no graphs at all: a capture that baked in another stream's workspace would otherwise be timed unnoticed).
            # Those validation replays run BACK TO BACK right in front of the warm-up (validate_graphs below), their frames are kept and
            # compared after the timed region: done one by one with a synchronisation each (as until round 6) they left the board idle for
            # seconds, and the --warmup steps (10 ms at W = 5) do not bring it back to its steady state — tools/region_warm_probe.py on one
            # box: the 20-step region 1.962-1.972 ms per step after 1 s of idle + 5 steps, 1.910-1.940 with 12 more steps in front.
            try:
                pools = [torch.cuda.graph_pool_handle() for _ in streams]
                for s_i, st in enumerate(streams):
                    for k in range(npairs):
                        gi = Hn.GraphedInterpolator(model, args, frames[k], t, pyramid=pyrs[k], stream=st, pool=pools[s_i], check="defer")
                        graphs[(s_i, k)] = gi
                        graph_outs[(s_i, k)] = gi.out
                torch.cuda.synchronize()
                graph_state.update(on=True, why="%d graphs (streams x pairs; fldr_harness.GraphedInterpolator) captured" % len(graphs))
            except Exception as e:                                   # eager steps still work
                graphs.clear(); graph_outs.clear()
                graph_state.update(on=False, why="capture failed: %r" % (e,))
                torch.cuda.synchronize()

        def validate_graphs():
            """Every graph replayed once, in the order the loop uses them, no synchronisation; frames kept for verify_graphs()."""
            order = []
            for i in range(len(streams) * npairs):                    # the loop's own order (consecutive replays on different streams) ...
                sk = (i % len(streams), i % npairs)
                if sk not in order:
                    order.append(sk)
            order += [sk for sk in graphs if sk not in order]         # ... then the combinations the loop never reaches
            for sk in order:
                graphs[sk].first_replay()

        def verify_graphs():
            try:
                for gi in graphs.values():
                    gi.verify()
                graph_state["why"] = ("%d graphs (streams x pairs; fldr_harness.GraphedInterpolator), every one replayed once right in front of the warm-up "
                                      "and == its eager frame bit for bit (compared after the timed region)" % len(graphs))
                return True
            except Exception as e:
                graphs.clear(); graph_outs.clear()
                graph_state.update(on=False, why="replay check failed: %r; the timed region was repeated with eager steps" % (e,))
                return False

        def step(i):
            if graph_state["on"]:
                s_i, k = i % len(streams), i % npairs
                return graphs[(s_i, k)].replay()
            return eager_step(i)
    else:
        def step(i):
            time.sleep(1e-3)
            return None
        eager_step = step

    out = None
    for i in range(max(len(streams) if gpu else 1, npairs)):      # prime every stream's allocator pool once (untimed)
        out = eager_step(i)
    sync()
    if gpu and not a.no_graphs:
        capture_graphs()
        if graph_state["on"]:
            validate_graphs()

    def timed_region():
        """--warmup untimed steps, then EXACTLY --steps steps between barrier + synchronize on both sides."""
        o = None
        for i in range(a.warmup):
            o = step(i)
        sync()
        barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(a.steps):
            o = step(i)
        sync()
        barrier()
        return time.perf_counter() - t0, o
    # (the single-stream latency legs run AFTER the timed region: the --warmup steps directly precede it)
    dt_local, out = timed_region()
    if gpu and graph_state["on"] and not verify_graphs():          # a replay that differs from its eager frame: nothing of it is reported
        dt_local, out = timed_region()
    # ---- sustained rate: the same loop for >= --sustained-s seconds (clocks settle, caches in steady state) ---
    dt = max_over_ranks(dt_local, device)
    if gpu:
        def one_at_a_time(fn, reps=6):                            # single-stream latency: a forward alone on the GPU, host waits for it
            for k in range(2):
                fn(k * len(streams)); sync()
            tl = time.perf_counter()
            for k in range(reps):
                fn(k * len(streams)); sync()                     # (multiples of the stream count: always stream 0)
            return (time.perf_counter() - tl) / reps * 1e3
        latency_ms = one_at_a_time(step)
        latency_eager_ms = one_at_a_time(eager_step) if graph_state["on"] else latency_ms
    if gpu and a.sustained_s > 0:
        n_s = max(a.steps, int(a.sustained_s / max(dt / a.steps, 1e-5)) + 1)      # the same step count on every rank
        barrier()
        sync()
        t1 = time.perf_counter()
        for i in range(n_s):
            out = step(i)
        sync()
        barrier()
        d_s = max_over_ranks(time.perf_counter() - t1, device)
        sustained = {"seconds": round(d_s, 3), "steps": n_s, "ms_per_step": round(d_s / n_s * 1e3, 3),
                     "value": round(world * n_s / d_s, 3)}
    if gpu and graph_state["on"]:                                  # the same loop enqueued eagerly, for the record (untimed for `value`)
        n_e = max(a.steps, 40)
        barrier()
        sync()
        t1 = time.perf_counter()
        for i in range(n_e):
            out = eager_step(i)
        t_host = time.perf_counter() - t1
        sync()
        barrier()
        graph_state["eager_ms_per_step"] = round((time.perf_counter() - t1) / n_e * 1e3, 3)
        graph_state["eager_host_enqueue_ms_per_step"] = round(t_host / n_e * 1e3, 3)
    if gpu:                                            # every rank: a finite frame, no activation outside the fp16 split's range
        import fldr_hip
        assert out.shape[-2:] == (a.height, a.width) and torch.isfinite(out).all()
        fldr_hip.check_range()
    per_rank = gather_floats(a.steps / dt_local, device)
    x_test = None
    if gpu and a.xtest_dir:                            # BASELINE metric, second half: PSNR on X-Test when the data is there (every rank: collective)
        x_test = Hn.evaluate_dir(a.xtest_dir, multiple=a.xtest_multiple, model=model, args=args, device=device, rank=rank, world=world)
    pg_world = dist.get_world_size() if use_pg else 1
    if use_pg:
        # Every collective of the run is behind us: release the other ranks NOW.  What follows on rank 0 (informational legs, roofline
        # of the dominant kernel, the CPU baseline: ~1 min) involves no other rank, so none of them waits in a barrier that a slow or
        # stuck leg would turn into a collective timeout over there.
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    if gpu:
        # end-to-end on the device (informational, single stream): uint8 frames -> ingest kernels (normalise, reflect
        # pad, bicubic pyramid) -> forward -> rounded uint8 frame (fldr_frame_metrics)
        u8 = Hn.synthetic_pair(a.height, a.width, seed=my_pairs[0]).unsqueeze(0).to(device)
        for _ in range(3):
            Hn.interpolate_u8(model, args, u8, t)
        sync()
        t1 = time.perf_counter()
        for _ in range(max(1, a.steps // 4)):
            Hn.interpolate_u8(model, args, u8, t)
        sync()
        dt_e2e = (time.perf_counter() - t1) / max(1, a.steps // 4)
        # BASELINE config 5 (informational; never `value`): the same loop with plain fp16 convolution inputs (one MFMA per
        # product instead of three; ~74 dB against the fp32-class frame, tests/test_gpu_parity.py::test_fp16_conv_path_config5)
        # Informational (never `value`): the same loop on pairs under a smoothly varying motion (zoom + rotation + shift,
        # fldr_harness.synthetic_pair_varying) instead of the global shift of the headline pairs — the scatter kernels'
        # cost depends on the flow field.
        varying = None
        if rank == 0 and a.varying_motion_steps > 0:
            with torch.no_grad():
                vf = [Hn.frames_from_uint8(Hn.synthetic_pair_varying(a.height, a.width, seed=100 + k)).to(device) for k in range(2)]
                vp = [Hn.build_pyramid(Hn.pad_frames(f, args), args) for f in vf]

            def vstep(i):
                with torch.cuda.stream(streams[i % len(streams)]), torch.no_grad():
                    return Hn.interpolate(model, args, vf[i % 2], t, pyramid=vp[i % 2])
            for i in range(2 * len(streams)):
                vout = vstep(i)
            sync()
            t1 = time.perf_counter()
            for i in range(a.varying_motion_steps):
                vout = vstep(i)
            sync()
            dv = time.perf_counter() - t1
            assert torch.isfinite(vout).all()
            varying = {"what": "the same loop on pairs under a smoothly varying motion field (1.2 % zoom + 0.25 deg rotation + shift; "
                               "2 distinct pairs): reported for reference only", "steps": a.varying_motion_steps,
                       "ms_per_step": round(dv / a.varying_motion_steps * 1e3, 3), "pairs_per_s_this_gpu": round(a.varying_motion_steps / dv, 2)}
            del vf, vp
        # Informational (never `value`): SURVEY 8d's unit of work — two uint8 frames resident on the GPU -> ingest kernels
        # (normalise, reflect pad, bicubic pyramid) -> forward -> rounded uint8 frame — with the same number of pairs in flight
        # on separate streams as the headline loop.
        incl = None
        if rank == 0 and a.incl_ingest_steps > 0:
            u8s = [Hn.synthetic_pair(a.height, a.width, seed=my_pairs[k % npairs]).unsqueeze(0).to(device) for k in range(npairs)]

            def istep(i):
                with torch.cuda.stream(streams[i % len(streams)]), torch.no_grad():
                    return Hn.interpolate_u8(model, args, u8s[i % npairs], t)[0]
            for i in range(2 * len(streams)):
                iout = istep(i)
            sync()
            t1 = time.perf_counter()
            for i in range(a.incl_ingest_steps):
                iout = istep(i)
            sync()
            di = time.perf_counter() - t1
            assert iout.dtype == torch.uint8
            incl = {"what": "uint8 frame pair resident in HBM -> ingest (normalise, reflect pad, bicubic pyramid) -> forward -> rounded uint8 "
                            "frame, %d pairs in flight: SURVEY 8d's unit of work; reported for reference only" % len(streams),
                    "steps": a.incl_ingest_steps, "ms_per_step": round(di / a.incl_ingest_steps * 1e3, 3),
                    "pairs_per_s_this_gpu": round(a.incl_ingest_steps / di, 2)}
            del u8s
        # Informational: BASELINE config 3 — 4096x2160 (padded 2304x4096), t = 1/8 ... 7/8 (7 output frames per pair, main.py:833)
        # with and without the pair-invariant cache; single stream, output frames per second.
        multi_t = None
        if rank == 0 and a.multi_t_pairs > 0 and (a.height, a.width) == (H4K, W4K):
            tvals = [k / 8.0 for k in range(1, 8)]
            with torch.no_grad():
                mf = [Hn.frames_from_uint8(Hn.synthetic_pair(2160, 4096, seed=200 + k)).to(device) for k in range(2)]
                mp = [Hn.build_pyramid(Hn.pad_frames(f, args), args) for f in mf]

                def pair_cached(i):
                    return Hn.interpolate_multi(model, args, mf[i % 2], tvals, pyramid=mp[i % 2])

                def pair_cached_streams(i):
                    return Hn.interpolate_multi(model, args, mf[i % 2], tvals, pyramid=mp[i % 2], streams=streams)

                def pair_plain(i):
                    return [Hn.interpolate(model, args, mf[i % 2], torch.full((1, 1), tv, device=device), pyramid=mp[i % 2]) for tv in tvals]
                rec = {}
                for name, fn in (("pair_cache", pair_cached), ("pair_cache_%d_streams" % len(streams), pair_cached_streams), ("no_cache", pair_plain)):
                    # two untimed pairs (the second one allocates the outputs that live beside the previous pair's), then every pair timed on
                    # its own: the rate is the MEDIAN pair's — twice in four full bench runs of round 6 one pair of this leg took 3-4x as long
                    # (a stall outside the kernels: the same loop alone never shows it, tools/multit_latency_probe.py) and the mean hid the
                    # other three; all pair times are reported
                    mo = fn(0)
                    mo = fn(1)
                    sync()
                    each = []
                    for i in range(a.multi_t_pairs):
                        t1 = time.perf_counter()
                        mo = fn(i)
                        sync()
                        each.append(time.perf_counter() - t1)
                    assert len(mo) == 7 and torch.isfinite(mo[-1]).all()
                    med = sorted(each)[len(each) // 2] if len(each) % 2 else 0.5 * (sorted(each)[len(each) // 2 - 1] + sorted(each)[len(each) // 2])
                    rec[name] = {"ms_per_pair": round(med * 1e3, 3), "output_frames_per_s": round(7 / med, 2), "ms_per_pair_each": [round(x * 1e3, 2) for x in each],
                                 "ms_per_pair_mean": round(sum(each) / len(each) * 1e3, 3)}
            multi_t = dict(rec, what="BASELINE config 3: 4096x2160 pair (padded 2304x4096), 7 outputs per pair (t = 1/8 ... 7/8), with / without "
                                     "the pair-invariant cache on one stream, and with the cache and the six later outputs dealt to the side streams; reported for reference only", pairs=a.multi_t_pairs)
            del mf, mp
        fp16_mode = None
        if rank == 0 and a.fp16_mode_steps > 0:
            prev = fldr_hip.CONV_PRECISION
            try:
                fldr_hip.CONV_PRECISION = "fp16"
                for i in range(npairs):
                    eager_step(i)                        # (eager: the captured graphs hold the split-precision forward)
                sync()
                t1 = time.perf_counter()
                for i in range(a.fp16_mode_steps):
                    eager_step(i)
                sync()
                d5 = time.perf_counter() - t1
                fp16_mode = {"what": "3x3 convolutions on plain fp16 inputs (FLDR_CONV_PRECISION=fp16, BASELINE config 5): not fp32-equivalent, "
                                     "reported for reference only", "steps": a.fp16_mode_steps, "ms_per_step": round(d5 / a.fp16_mode_steps * 1e3, 3),
                             "pairs_per_s_this_gpu": round(a.fp16_mode_steps / d5, 2)}
            finally:
                fldr_hip.CONV_PRECISION = prev
        # BASELINE config 5 proper (informational; never `value`): Xiph-4K geometry — a 4096x2160 pair (padded 2304x4096), t = 0.5,
        # fp16-input MFMA convolutions, pairs in flight as in the headline loop — with the roofline of its dominant kernel and of the path
        # (fp16 convolutions make the path HBM-bound by the survey's model: 5.636 GB -> 0.705 ms), and the PSNR of its rounded frame against
        # the fp32-class (split) frame of the same pair.
        config5 = None
        if rank == 0 and a.config5_steps > 0 and (a.height, a.width) == (H4K, W4K):
            prev = fldr_hip.CONV_PRECISION
            try:
                with torch.no_grad():
                    xf = [Hn.frames_from_uint8(Hn.synthetic_pair(2160, 4096, seed=300 + k)).to(device) for k in range(max(2, len(streams)))]
                    xp_ = [Hn.build_pyramid(Hn.pad_frames(f, args), args) for f in xf]
                    ref5 = Hn.interpolate(model, args, xf[0], t, pyramid=xp_[0])
                fldr_hip.CONV_PRECISION = "fp16"

                def xstep(i):
                    k = i % len(xf)
                    with torch.cuda.stream(streams[i % len(streams)]), torch.no_grad():
                        return Hn.interpolate(model, args, xf[k], t, pyramid=xp_[k])
                for i in range(2 * len(streams)):
                    xo = xstep(i)
                sync()
                t1 = time.perf_counter()
                for i in range(a.config5_steps):
                    xo = xstep(i)
                sync()
                d5x = time.perf_counter() - t1
                with torch.no_grad():
                    o5 = Hn.interpolate(model, args, xf[0], t, pyramid=xp_[0])
                sync()
                ms5 = d5x / a.config5_steps * 1e3
                pm5 = PATH_MODEL[(2160, 4096)]
                fl_hbm = pm5["hbm_bytes"] / (PEAK_HBM_GBPS * 1e9) * 1e3
                fl_mfma = pm5["flops"] / (PEAK_FP16_MFMA_TFLOPS * 1e12) * 1e3
                config5 = {"what": "BASELINE config 5: 4096x2160 pair (padded 2304x4096), t = 0.5, 3x3 convolutions on plain fp16 inputs "
                                   "(FLDR_CONV_PRECISION=fp16: one MFMA per product; not fp32-equivalent), %d pairs in flight; reported for reference only" % len(streams),
                           "steps": a.config5_steps, "ms_per_step": round(ms5, 3), "pairs_per_s_this_gpu": round(a.config5_steps / d5x, 2),
                           "psnr_8bit_vs_split_path_db": Hn.psnr(Hn.to_uint8_image(ref5[0]), Hn.to_uint8_image(o5[0])),
                           "max_abs_diff_vs_split_path": float((o5 - ref5).abs().max()),
                           "roofline": dict(dominant_conv_roofline(model, 2304 // 8, 4096 // 8, device, a.steps, precision="fp16"),
                                            path={"what": "whole forward against its binding floor: compulsory HBM traffic of SURVEY App. C (5.636 GB at 4096x2160) at 8 TB/s; "
                                                          "the fp16 matrix work is the smaller term",
                                                  "hbm_bytes": pm5["hbm_bytes"], "hbm_floor_ms": round(fl_hbm, 3), "mfma_floor_ms_fp16": round(fl_mfma, 3),
                                                  "ms_per_step": round(ms5, 3), "frac": round(max(fl_hbm, fl_mfma) / ms5, 4),
                                                  "achieved_GBps": round(pm5["hbm_bytes"] / (ms5 * 1e-3) / 1e9, 1)})}
                del xf, xp_, ref5, o5
            finally:
                fldr_hip.CONV_PRECISION = prev
    if rank == 0:
        hp = ((a.height + 255) // 256 * 256, (a.width + 255) // 256 * 256)
        res = {
            "metric": "4K frame-pairs interpolated/sec", "value": round(world * a.steps / dt, 3), "unit": "frame-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (3x3 convs: 3 x fp16-split MFMA with fp32 accumulation — error against fp64 within 1.5x (mean) / 2x (max) of the exact fp32-MFMA "
                     "kernel's on every layer shape, tests/test_gpu_parity.py::test_split_fp16_conv_is_fp32_equivalent; rest fp32/fp64)",
            "data": "synthetic",
            "config": {"workload": "single %dx%d frame pair per step (padded %dx%d), fLDRnet 5-scale test path "
                                   "(--papermodel --test5scales), t=0.5, shipped checkpoint weights, fp64 output frame; "
                                   "%d distinct pairs per GPU rotated through the loop" % (a.width, a.height, hp[0], hp[1], npairs),
                       "parallelism": "dp%d (independent pairs, no data-path collective)" % world,
                       "pairs_in_flight": a.streams, "distinct_pairs_per_gpu": npairs,
                       "per_rank_pairs_per_s": [round(x, 2) for x in per_rank], "setup_threads_per_rank": setup_threads},
        }
        if gpu:
            res["config"]["hip_graphs"] = dict(graph_state)
        if use_pg:
            res["config"]["process_group"] = {"backend": pg_backend + (" (RCCL)" if pg_backend == "nccl" else ""), "world_size": pg_world,
                                              "forced_at_world_size_1": world == 1}
        if a.share_gpu and world > 1:
            res["rehearsal"] = ("%d ranks SHARE device 0 (--share-gpu): the multi-process launch, sharding, barriers and reductions around real HIP work on a "
                                "1-GPU box; `value` is NOT a measurement of %d GPUs" % (world, world))
        if a.dry:
            res["dry"] = True
            res["data"] = "none (dry run: 1 ms sleep per step)"
            if not a.no_cpu_baseline:                  # the leg itself runs in a dry launch too (rank 0, any world size), on a small pair
                import fldr_harness as Hn
                fr = Hn.frames_from_uint8(Hn.synthetic_pair(256, 256, seed=0))
                res["cpu_baseline"], _ = cpu_baseline(fr, torch.tensor([[0.5]]), None, runs=1)
                res["cpu_baseline"]["unit"] = "256x256 frame-pairs/s (dry run)"
        if gpu:
            res["config"].update({"pyramid_bytes_per_pair": pyr_bytes, "single_stream_latency_ms": round(latency_ms, 3), "single_stream_latency_eager_ms": round(latency_eager_ms, 3),
                                  "ms_uint8_in_to_uint8_out_single_stream": round(dt_e2e * 1e3, 3)})
            if sustained:
                res["sustained"] = sustained
            if fp16_mode:
                res["fp16_conv_mode"] = fp16_mode
            if config5:
                res["config5"] = config5
            if varying:
                res["varying_motion"] = varying
            if incl:
                res["incl_ingest"] = incl
            if multi_t:
                res["multi_t"] = multi_t
            res["roofline"] = dominant_conv_roofline(model, hp[0] // 8, hp[1] // 8, device, a.steps)
            pm = PATH_MODEL.get((a.height, a.width))
            if pm:
                ms_step = dt / a.steps * 1e3
                floor_hbm = pm["hbm_bytes"] / (PEAK_HBM_GBPS * 1e9) * 1e3
                floor_mfma = 3 * pm["flops"] / (PEAK_FP16_MFMA_TFLOPS * 1e12) * 1e3
                res["roofline"]["path"] = {
                    "what": "whole forward against its binding floor: compulsory HBM traffic of SURVEY App. C at 8 TB/s (the 3 x fp16-split "
                            "matrix work is the smaller term)", "hbm_bytes": pm["hbm_bytes"], "hbm_floor_ms": round(floor_hbm, 3),
                    "mfma_floor_ms_split_fp16": round(floor_mfma, 3), "fp32_mfma_floor_ms": round(pm["flops"] / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3, 3),
                    "ms_per_step": round(ms_step, 3), "frac": round(max(floor_hbm, floor_mfma) / ms_step, 4),
                    "frac_single_stream": round(max(floor_hbm, floor_mfma) / latency_ms, 4),
                    "achieved_GBps": round(pm["hbm_bytes"] / (ms_step * 1e-3) / 1e9, 1)}
                if (a.height, a.width) == (H4K, W4K):
                    mv, mv_src = _moved_bytes_per_forward()
                    if mv:
                        res["roofline"]["path"].update({"moved_bytes_per_forward": mv, "moved_over_algorithmic": round(mv / pm["hbm_bytes"], 3),
                                                        "moved_bytes_source": mv_src})
                    en = _energy_record()
                    if en:
                        res["roofline"]["path"]["energy"] = en
            if not a.no_cpu_baseline:                  # rank 0, any world size (3 timed forwards at one GPU, 1 beyond: `runs`)
                with torch.no_grad():
                    g0 = Hn.interpolate(model, args, frames[0], t, pyramid=pyrs[0]).cpu()
                res["cpu_baseline"], res["parity"] = cpu_baseline(frames[0].cpu(), t.cpu(), g0, runs=3 if world == 1 else 1)
            if x_test is not None:
                res["x_test"] = dict(x_test, dir=a.xtest_dir, multiple=a.xtest_multiple)
                res.setdefault("parity", {})["x_test_psnr"] = x_test["psnr"]
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
