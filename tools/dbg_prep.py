"""level0_prep at the 4K shape with a smooth synthetic flow: us per call (LIB=path selects an experimental build)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
if os.environ.get("LIB"): hip.LIB_PATH = os.environ["LIB"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W, h, w = 2304, 3840, 288, 480
x = torch.rand(1, 3, 2, H, W, device=dev) * 2 - 1
lo = torch.nn.functional.interpolate(torch.randn(1, 4, 9, 15, device=dev) * 1.5, size=(h, w), mode="bilinear").contiguous()
t = torch.tensor([[0.5]], device=dev)
run = lambda: hip.level0_prep(lo, x[:, :, 0], x[:, :, 1], t, H, W, -1.9, -1.8, withmask=True, want_z=True)
for _ in range(3): r = run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): r = run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("LIB", "product"), "level0_prep %.1f us" % (e0.elapsed_time(e1) / 10 * 1e3), "checksum %.6f" % sum(v.double().mean().item() for k, v in r.items() if k != "_keep"), flush=True)
