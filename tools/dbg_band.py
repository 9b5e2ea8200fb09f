"""Image band splat (atomic-free) at the 4K shape with a smooth +-8 px flow: us per launch.  FLDR_LIB selects a variant build."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W = 2304, 3840
sets = []
for k in range(3):
    lo = (torch.rand(1, 2, H // 8, W // 8, device=dev) - 0.5) * 4 + torch.tensor([6.0, 4.0], device=dev).view(1, 2, 1, 1)
    flow = hip.resize_bilinear(hip.resize_bilinear(lo, H // 2, W // 2), H, W)
    sets.append((torch.rand(1, 3, H, W, device=dev) * 2 - 1, flow, torch.rand(1, 1, H, W, device=dev) * -2))
def timeit(fn, n=16):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(os.environ.get("FLDR_LIB", "product").split("/")[-1], "band splat (bounds + band): %.1f us" % timeit(lambda i: hip.softsplat_fused(*sets[i % 3], "softmax", kernel="tile")), flush=True)
