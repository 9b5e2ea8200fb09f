#!/usr/bin/env python3
"""Run ONE kernel of the 4K forward back to back for a few seconds (tools/power_by_kernel.sh samples rocm-smi meanwhile):
    python tools/kernel_loop.py <conv96|conv_dec2|prep|enc1|dec3|splat|pca|fsplat|idle> [seconds]
Prints the average time per call."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip, fldr_harness as Hn
which = sys.argv[1]; secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, W = 2304, 3840
model, _, args = Hn.prepare_model(dev)
un = model.vfinet.refine_unet
if which == "idle":
    fn = None
elif which.startswith("conv96"):
    # conv96 | conv96:nc4 (four consumer waves) | conv96:r32 (the 32x32x16 kernel forced): the variants through the test build's hooks
    if ":" in which:
        L = hip.enter_test_hooks()
        if which.endswith(":nc4"): L.fldr_debug_ring_consumers(4)
        if which.endswith(":r32"): L.fldr_debug_ring32(2)
    xs = [hip.spk_pack(torch.rand(1, 96, 288, 480, device=dev) * 2 - 1) for _ in range(4)]
    c = model.rec_ctx_ds[0]
    fn = lambda i: hip.conv2d_spk([xs[i % 4]], c.weight, c.bias, relu=True, want_f32=False, want_spk=True)
elif which == "conv_dec2":
    a = [hip.spk_pack(torch.rand(1, 32, 576, 960, device=dev)) for _ in range(2)]; b = [hip.spk_pack(torch.rand(1, 16, 1152, 1920, device=dev)) for _ in range(2)]
    fn = lambda i: hip.conv2d_spk([a[i % 2], b[i % 2]], un.dec2.weight, un.dec2.bias, relu=True, up2=[True, False], want_f32=False, want_spk=True)
elif which == "prep":
    lo = torch.tensor([-0.75, -0.5, 0.75, 0.5], device=dev).view(1, 4, 1, 1) + torch.randn(1, 4, 288, 480, device=dev) * 0.02
    fr = [torch.rand(1, 3, 2, H, W, device=dev) * 2 - 1 for _ in range(3)]; t4 = torch.tensor([0.5], device=dev).view(1, 1, 1, 1)
    fn = lambda i: hip.level0_prep(lo, fr[i % 3][:, :, 0], fr[i % 3][:, :, 1], t4, H, W, -1.9, -1.9, withmask=True, want_z=True)
elif which == "enc1":
    xs = [torch.rand(1, 26, H, W, device=dev) * 2 - 1 for _ in range(3)]
    fn = lambda i: hip.conv2d([xs[i % 3]], un.enc1.weight, un.enc1.bias, stride=2, relu=True, want_f32=False, want_spk=True)
elif which == "dec3":
    d2 = [hip.spk_pack(torch.rand(1, 16, H // 2, W // 2, device=dev)) for _ in range(2)]
    cands = [[torch.rand(1, 3, H, W, device=dev) * 2 - 1 for _ in range(6)] for _ in range(2)]; t = torch.tensor([[0.5]], device=dev)
    fn = lambda i: hip.dec3_synth(d2[i % 2], un.dec3.weight, un.dec3.bias, cands[i % 2], t, 1.5616)
elif which == "splat":
    fr = [torch.rand(1, 3, 2, H, W, device=dev) * 2 - 1 for _ in range(3)]
    lo = torch.tensor([-0.75, -0.5, 0.75, 0.5], device=dev).view(1, 4, 1, 1) + torch.randn(1, 4, 288, 480, device=dev) * 0.02
    t4 = torch.tensor([0.5], device=dev).view(1, 1, 1, 1)
    r = hip.level0_prep(lo, fr[0][:, :, 0], fr[0][:, :, 1], t4, H, W, -1.9, -1.9, withmask=True, want_z=True)
    bw = hip.splat_bounds_upsampled_pair(lo, t4, "images", 8, H, W)
    fn = lambda i: hip.softsplat_acc64([fr[i % 3][:, :, 0], fr[i % 3][:, :, 1]], [r["flow_t0"], r["flow_t1"]], [r["z0"], r["z1"]], "softmax", bounds_ws=bw)
elif which == "pca":
    pyrs = [[(torch.rand(6, H >> l, W >> l, device=dev) * 2 - 1) for l in range(6)] for _ in range(3)]
    ev, mean, mv = model.EV8.detach(), model.Mean8.detach(), model.meanVec8.detach()
    fn = lambda i: hip.pca_project_pyramid(pyrs[i % 3], ev, mean, mv, want_f32=True, want_spk=True)
else:
    raise SystemExit("unknown kernel " + which)
if fn is None:
    print("START idle", flush=True); time.sleep(secs); print("END idle 0  idle %.1f s" % secs); sys.exit(0)
for i in range(5): fn(i)
torch.cuda.synchronize()
n = 0; t0 = time.perf_counter()
print("START %s %.3f" % (which, time.time()), flush=True)
while time.perf_counter() - t0 < secs:
    for i in range(20): fn(n + i)
    n += 20
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("END %s %.3f  %.1f us per call (%d calls)" % (which, time.time(), dt / n * 1e6, n), flush=True)
