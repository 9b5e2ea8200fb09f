"""Host-side enqueue time per forward vs GPU time (are we launch-bound?)."""
import os, sys, time, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
frames = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=0)).to(dev)
t = torch.tensor([[0.5]], device=dev)
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    def step(i):
        with torch.cuda.stream(streams[i % 3]):
            return Hn.interpolate(model, args, frames, t, pyramid=pyr)
    for i in range(6): step(i)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("host enqueue %.3f ms per forward; total %.3f ms per forward (GPU-bound if total >> enqueue)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
