"""pca_project_stream at the 4K level-0 shape: us per call (LIB=path selects an experimental build)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
if os.environ.get("LIB"): hip.LIB_PATH = os.environ["LIB"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
ev = torch.randn(16, 64, device=dev, dtype=torch.float64); mean = torch.randn(64, device=dev, dtype=torch.float64) * 0.1
mv = torch.rand(16, device=dev, dtype=torch.float64) + 0.5
planes = torch.rand(6, 2304, 3840, device=dev) * 2 - 1
for _ in range(3): r = hip.pca_project_stream(planes, ev, mean, mv, want_spk=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): r = hip.pca_project_stream(planes, ev, mean, mv, want_spk=True)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("LIB", "product"), "pca stream (init + project + rescale) %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), "checksum %.9f" % r[0].double().mean().item(), flush=True)
