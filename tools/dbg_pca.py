"""PCA projection of a whole 4K pyramid: per-level one-pass kernels (fldr_pca_project_stream x 6) vs the two-launch
pyramid kernels (fldr_pca_project_pyramid): us per forward, with the input rotating over NP distinct pyramids (so that
the level-0 planes do not sit in the Infinity Cache) and with one resident pyramid."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
if os.environ.get("LIB"): hip.LIB_PATH = os.environ["LIB"]
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
ev, mean, mv = model.EV8.detach(), model.Mean8.detach(), model.meanVec8.detach()
H, W = int(os.environ.get("FH", 2304)), int(os.environ.get("FW", 3840))
NP = 3
pyrs = [[(torch.rand(6, H >> l, W >> l, device=dev) * 2 - 1) for l in range(6)] for _ in range(NP)]
def t(fn, n=20):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
old = lambda i: [hip.pca_project_stream(p, ev, mean, mv, want_spk=True) for p in pyrs[i % NP]]
new = lambda i: hip.pca_project_pyramid(pyrs[i % NP], ev, mean, mv, want_f32=True, want_spk=True)
old1 = lambda i: [hip.pca_project_stream(p, ev, mean, mv, want_spk=True) for p in pyrs[0]]
new1 = lambda i: hip.pca_project_pyramid(pyrs[0], ev, mean, mv, want_f32=True, want_spk=True)
l0o = lambda i: hip.pca_project_stream(pyrs[i % NP][0], ev, mean, mv, want_spk=True)
l0n = lambda i: hip.pca_project_pyramid(pyrs[i % NP][:1], ev, mean, mv, want_f32=True, want_spk=True)
print("%dx%d pyramid, rotating inputs : per-level %.1f us, pyramid kernels %.1f us" % (H, W, t(old), t(new)))
print("%dx%d pyramid, resident input  : per-level %.1f us, pyramid kernels %.1f us" % (H, W, t(old1), t(new1)))
print("level 0 only, rotating inputs   : per-level %.1f us, pyramid kernels %.1f us" % (t(l0o), t(l0n)))
