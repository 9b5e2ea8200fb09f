"""Single-stream latency of one 4K forward: eager launches vs one hipGraph replay; small-level sub-groups on / off."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn, fldr_hip as hip
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
frames = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=0)).to(dev)
t = torch.tensor([[0.5]], device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)
    for su in (-1, 96):
        hip.lib().fldr_debug_spk_small_units(su)
        ref = Hn.interpolate(model, args, frames, t, pyramid=pyr).clone()
        print("small_units", su, "eager single-stream ms", round(timeit(lambda: Hn.interpolate(model, args, frames, t, pyramid=pyr)), 3), flush=True)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): Hn.interpolate(model, args, frames, t, pyramid=pyr)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            o = Hn.interpolate(model, args, frames, t, pyramid=pyr)
        torch.cuda.synchronize()
        print("small_units", su, "graph replay ms", round(timeit(g.replay), 3), "err", (o - ref).abs().max().item(), flush=True)
