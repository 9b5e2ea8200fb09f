"""The uint8-in -> uint8-out leg (ingest + forward + rounded frame; 3 pairs in flight on 3 streams, eager launches, as bench.py's incl_ingest)
with the 8-bit frame taken straight from the synthesis kernel against fp64 frame + fldr_frame_metrics, alternating on one box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import torch
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
H, W = 2160, 3840
u8s = [Hn.synthetic_pair(H, W, seed=k).unsqueeze(0).to(dev) for k in range(4)]
t = torch.tensor([[0.5]], device=dev)
streams = [torch.cuda.Stream() for _ in range(3)]
def step(i):
    with torch.cuda.stream(streams[i % 3]), torch.no_grad():
        return Hn.interpolate_u8(model, args, u8s[i % 4], t)[0]
def run(n=60):
    for i in range(6): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for direct in (True, False):
        Hn.U8_DIRECT = direct
        print("8-bit frame %s: %.3f ms per pair" % ("from the synthesis kernel" if direct else "fp64 frame + frame_metrics", run()), flush=True)
