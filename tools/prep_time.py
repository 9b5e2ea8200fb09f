"""level0_prep at the 4K shape, rotating inputs: microseconds per call (HIP events), for A/B of library variants (FLDR_LIB)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import torch
import fldr_hip as hip
dev = torch.device("cuda:0")
H, W, up = 2304, 3840, 8
lo = [torch.tensor([-0.75, -0.5, 0.75, 0.5], device=dev).view(1, 4, 1, 1) * (1 + k) + torch.randn(1, 4, H // up, W // up, device=dev) * 0.5 for k in range(3)]
frames = [torch.rand(1, 3, 2, H, W, device=dev) * 2 - 1 for _ in range(3)]
t4 = torch.tensor([0.5], device=dev).view(1, 1, 1, 1)
def run(i): return hip.level0_prep(lo[i % 3], frames[i % 3][:, :, 0], frames[i % 3][:, :, 1], t4, H, W, 20.0, 20.0, withmask=True, want_z=True)
for i in range(6): run(i)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(60): r = run(i)
    e1.record(); torch.cuda.synchronize()
    print("level0_prep: %.1f us per call" % (e0.elapsed_time(e1) / 60 * 1e3), flush=True)
