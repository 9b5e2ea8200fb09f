"""dec3_synth at the 4K shape: us per launch (LIB=path selects an experimental build)."""
import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip
if os.environ.get("LIB"): hip.LIB_PATH = os.environ["LIB"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W = 2304, 3840
d2 = torch.randn(1, 16, H // 2, W // 2, device=dev)
w3 = torch.randn(6, 16, 3, 3, device=dev) / 12
b3 = torch.randn(6, device=dev)
x = torch.rand(1, 3, 2, H, W, device=dev)
cands = [torch.rand(1, 3, H, W, device=dev) for _ in range(4)] + [x[:, :, 0], x[:, :, 1]]
t = torch.tensor([[0.5]], device=dev)
def run(): return hip.dec3_synth(d2, w3, b3, cands, t, 1.56)
for _ in range(3): o = run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): o = run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("LIB", "product"), "dec3_synth %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), "checksum %.9f" % o.double().mean().item(), flush=True)
