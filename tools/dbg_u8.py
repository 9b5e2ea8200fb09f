import os, sys, time, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
u8 = Hn.synthetic_pair(2160, 3840, seed=0).unsqueeze(0).to(dev)
t = torch.tensor([[0.5]], device=dev)
with torch.no_grad():
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Hn.interpolate_u8(model, args, u8, t)
        torch.cuda.synchronize(); print("iter %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
