import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
ev = torch.randn(16, 64, device=dev, dtype=torch.float64); mean = torch.randn(64, device=dev, dtype=torch.float64) * 0.1
mv = torch.rand(16, device=dev, dtype=torch.float64) + 0.5
planes = torch.rand(6, 2304, 3840, device=dev) * 2 - 1
for _ in range(5):
    hip.pca_project_stream(planes, ev, mean, mv, want_spk=True)
torch.cuda.synchronize()
