"""Do an MFMA-bound convolution and an HBM-bound level-0 kernel overlap when launched on two streams?  Times N launches of
each alone and both together (ring conv with 8 / 4 consumer waves, i.e. 3 / 2 waves per SIMD)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.enter_test_hooks()          # variant / tuning hooks: the test build (libfldr_hip_test.so)
dev = torch.device("cuda:0"); torch.manual_seed(0); L = hip.lib()
H, W = 2304, 3840
x = torch.rand(1, 96, 288, 480, device=dev); xp = hip.spk_pack(x); wt = torch.randn(96, 96, 3, 3, device=dev) / 30
flow = (torch.rand(1, 4, 288, 480, device=dev) - 0.5) * 2
fr = torch.rand(1, 3, 2, H, W, device=dev) * 2 - 1
t = torch.tensor([[0.5]], device=dev)
d2 = torch.rand(1, 16, H // 2, W // 2, device=dev); cands = [torch.rand(1, 3, H, W, device=dev) for _ in range(6)]
w3 = torch.randn(6, 16, 3, 3, device=dev) / 6; b3 = torch.randn(6, device=dev)
srcs26 = torch.rand(1, 26, H, W, device=dev); w4 = torch.randn(16, 26, 4, 4, device=dev) / 20
conv = lambda: hip.conv2d_spk([xp], wt, None, relu=True, want_f32=False, want_spk=True)
others = {"prep": lambda: hip.level0_prep(flow, fr[:, :, 0], fr[:, :, 1], t, H, W, -1.9, -1.9),
          "dec3": lambda: hip.dec3_synth(d2, w3, b3, cands, t, 1.56),
          "enc1": lambda: hip.conv2d([srcs26], w4, None, stride=2, relu=True, precision="split", want_f32=True, want_spk=True)}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(fa, na, fb, nb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if fa:
        with torch.cuda.stream(sa):
            for _ in range(na): fa()
    if fb:
        with torch.cuda.stream(sb):
            for _ in range(nb): fb()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for cons in (8, 4):
    L.fldr_debug_ring_consumers(cons)
    for name, f in others.items():
        for _ in range(2): conv(); f()
        n_conv = 40
        ta = run(conv, n_conv, None, 0)
        n_o = max(2, int(ta / (run(None, 0, f, 4) / 4)))
        tb = run(None, 0, f, n_o)
        tab = run(conv, n_conv, f, n_o)
        print("ring%d + %-4s: conv x%d alone %.2f ms, %s x%d alone %.2f ms, together %.2f ms (sum %.2f, max %.2f)" % (cons, name, n_conv, ta, name, n_o, tb, tab, ta + tb, max(ta, tb)), flush=True)
