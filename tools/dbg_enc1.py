"""enc1 (26 -> 16, 4x4 stride 2 @2304x3840) through the split kernel: us per launch (LIB=path: experimental build)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
if os.environ.get("LIB"): hip.LIB_PATH = os.environ["LIB"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
parts = [3, 3, 3, 3, 2, 2, 2, 2, 3, 3]
srcs = [torch.randn(1, c, 2304, 3840, device=dev) for c in parts]
wt = torch.randn(16, 26, 4, 4, device=dev) / 20
b = torch.randn(16, device=dev)
run = lambda: hip.conv2d(srcs, wt, b, stride=2, relu=True, precision="split", want_spk=True)
for _ in range(3): o = run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): o = run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("LIB", "product"), "enc1 %.1f us" % (e0.elapsed_time(e1) / 10 * 1e3), "checksum %.6f" % o[0].double().mean().item(), flush=True)
