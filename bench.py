#!/usr/bin/env python3
"""Headline benchmark: 4K (3840x2160) frame pairs interpolated per second (BASELINE.json metric, config 2:
single 4K frame pair, 5-scale test path, t = 0.5) through the MI355X-native hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one (frame pair, t) forward = the reference's `model_net(input_gpuList, t, normInput=pyramid, ...)` call
(main.py:867) with the 6-level pyramid already resident in HBM.  Frame pairs are independent, so N ranks each
interpolate their own pair (weak scaling, no data-path collective); the only RCCL traffic is the barrier and a
max-reduction of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fldr-vfi_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H4K, W4K = 2160, 3840
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_FP16_MFMA_TFLOPS = 2500.0      # dense fp16/bf16 matrix peak
PEAK_HBM_GBPS = 8000.0


from fldr_harness import shard_pairs, host_cores, max_over_ranks  # noqa: E402


def dominant_conv_roofline(model, pyr, steps):
    """Roofline of the dominant kernel: the 96->96 3x3 convolution at the level-0 feature map (rec_ctx_ds.0/.2,
    conv_flow2.0/.2 launch this instance of conv3x3_spk_kernel; the 3x3 convolutions are ~30 % of the GPU time).  It
    runs on the fp16 matrix cores with the 3 x fp16 split (fp32-equivalent accuracy), i.e. it issues THREE fp16 MFMA
    flops per algorithmic flop.  algorithmic FLOPs per launch = 2 * cin * cout * 9 * pixels; achieved = 3 x that /
    launch time, against the dense fp16 MFMA peak.  Duration measured with HIP events on the launch stream, operands
    split-packed in HBM as inside the model.  The exact fp32-MFMA kernel of the same layer (FLDR_CONV_PRECISION=fp32)
    is timed beside it."""
    import fldr_hip
    h, w = pyr[0].shape[3] // 8, pyr[0].shape[4] // 8
    x = torch.rand(1, 96, h, w, device=pyr[0].device) * 2 - 1
    xp = fldr_hip.spk_pack(x)
    conv = model.rec_ctx_ds[0]
    out = torch.empty(1, 96, h, w, device=x.device)
    n = max(10, steps)

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

    ms = timed(lambda: fldr_hip.conv2d_spk([xp], conv.weight, conv.bias, relu=True, want_f32=False, want_spk=True))
    ms32 = timed(lambda: fldr_hip.conv2d([x], conv.weight, conv.bias, relu=True, out=out, precision="fp32"))
    flops = 2.0 * 96 * 96 * 9 * h * w
    ach = 3.0 * flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv3x3_spk_kernel<3,3> (3x3 96->96 @%dx%d, persistent workgroups, split-packed operands, "
                                       "3 x fp16-split v_mfma_f32_16x16x32_f16)" % (h, w),
            "achieved": round(ach, 1), "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP16_MFMA_TFLOPS, 4),
            "traffic": _measured_traffic(), "launch_ms": round(ms, 4), "flops_per_launch": flops,
            "issued_mfma_flops_per_launch": 3.0 * flops, "algorithmic_tflops": round(flops / (ms * 1e-3) / 1e12, 1),
            "exact_fp32_mfma_kernel": {"launch_ms": round(ms32, 4), "achieved": round(flops / (ms32 * 1e-3) / 1e12, 1),
                                       "peak": PEAK_FP32_MFMA_TFLOPS, "frac": round(flops / (ms32 * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}}


def _measured_traffic():
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); None when no measurement is on file."""
    p = os.path.join(ROOT, "profiles", "r01_conv96_spk_traffic.json")
    try:
        return json.load(open(p))["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(frames_cpu, t_cpu):
    """The oracle (CPU restatement of the reference, kind 'port') timed on this box's host cores: ONE 4K
    frame-pair forward (pyramid excluded, like the GPU number)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fldr_oracle as O
    import fldr_harness as Hn
    cores = host_cores()
    torch.set_num_threads(cores)
    w = O.load_weights(Hn.DEFAULT_WEIGHTS)
    pyr = O.pad_and_pyramid(frames_cpu)
    with torch.no_grad():
        t0 = time.time()
        O.forward(w, pyr, t_cpu)
        dt = time.time() - t0
    return {"value": round(1.0 / dt, 5), "unit": "4K frame-pairs/s", "cores": cores, "kind": "port",
            "sample": "1 forward of the same 3840x2160 pair (seed 0, t=0.5), oracle/fldr_oracle.py on torch-CPU, %.1f s" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent frame-pair forwards kept in flight on separate HIP streams (1 = strictly serial)")
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--width", type=int, default=W4K)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    import fldr_harness as Hn
    model, _, args = Hn.prepare_model(device)
    pair = shard_pairs(world, rank, world)[0]                    # one pair per rank, seed = pair index
    frames = Hn.frames_from_uint8(Hn.synthetic_pair(a.height, a.width, seed=pair)).to(device)
    t = torch.tensor([[0.5]], device=device)
    with torch.no_grad():
        pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)

        # Frame pairs are independent, so a serving loop keeps several in flight: step i is enqueued on HIP stream
        # i % streams.  Every step is a complete forward; overlapping them fills the CUs that the latency-bound
        # coarse pyramid levels (9x15 ... 72x120 feature maps) leave idle.
        streams = [torch.cuda.Stream(device=device) for _ in range(max(1, a.streams))]
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream())

        def step(i):
            with torch.cuda.stream(streams[i % len(streams)]):
                return Hn.interpolate(model, args, frames, t, pyramid=pyr)

        for i in range(len(streams)):          # prime every stream's allocator pool once (untimed, before the W warm-ups)
            out = step(i)
        torch.cuda.synchronize()
        for i in range(a.warmup):
            out = step(i)
        torch.cuda.synchronize()
        # single-stream latency of one forward (informational)
        tl = time.perf_counter()
        for _ in range(3):
            Hn.interpolate(model, args, frames, t, pyramid=pyr)
        torch.cuda.synchronize()
        latency_ms = (time.perf_counter() - tl) / 3 * 1e3
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            out = step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        # end-to-end on the device (informational, single stream): uint8 frames -> ingest kernels (normalise, reflect
        # pad, bicubic pyramid) -> forward -> rounded uint8 frame (fldr_frame_metrics)
        u8 = Hn.synthetic_pair(a.height, a.width, seed=pair).unsqueeze(0).to(device)
        for _ in range(3):                      # this path runs on the default stream: let its allocator pool settle first
            Hn.interpolate_u8(model, args, u8, t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(max(1, a.steps // 4)):
            Hn.interpolate_u8(model, args, u8, t)
        torch.cuda.synchronize()
        dt_e2e = (time.perf_counter() - t1) / max(1, a.steps // 4)
    assert out.shape[-2:] == (a.height, a.width) and torch.isfinite(out).all()
    dt = max_over_ranks(dt, device)

    if rank == 0:
        res = {
            "metric": "4K frame-pairs interpolated/sec", "value": round(world * a.steps / dt, 3), "unit": "frame-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (3x3 convs: 3 x fp16-split MFMA with fp32 accumulation, error <= exact fp32 MFMA; rest fp32/fp64)", "data": "synthetic",
            "config": {"workload": "single %dx%d frame pair per GPU (padded 2304x3840), fLDRnet 5-scale test path "
                                   "(--papermodel --test5scales), t=0.5, shipped checkpoint weights, fp64 output frame"
                                   % (a.width, a.height),
                       "parallelism": "dp%d (independent pairs, no data-path collective)" % world,
                       "pairs_in_flight": len(streams), "single_stream_latency_ms": round(latency_ms, 3),
                       "ms_uint8_in_to_uint8_out_single_stream": round(dt_e2e * 1e3, 3)},
        }
        res["roofline"] = dominant_conv_roofline(model, pyr, a.steps)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(frames.cpu(), t.cpu())
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
