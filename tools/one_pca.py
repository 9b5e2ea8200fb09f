"""REPS pyramid-PCA calls at the 4K level-0 shape (and the whole pyramid) for rocprofv3; LIB=<path> selects an experimental library."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
if os.environ.get("LIB"): hip.LIB_PATH = os.environ["LIB"]
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
ev, mean, mv = model.EV8.detach(), model.Mean8.detach(), model.meanVec8.detach()
NP = 3
pyrs = [[(torch.rand(6, 2304 >> l, 3840 >> l, device=dev) * 2 - 1) for l in range(6)] for _ in range(NP)]
for i in range(int(os.environ.get("REPS", 9))):
    hip.pca_project_pyramid(pyrs[i % NP][:1] if os.environ.get("L0") else pyrs[i % NP], ev, mean, mv, want_f32=True, want_spk=True)
torch.cuda.synchronize()
