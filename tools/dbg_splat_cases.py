import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W = 40, 200
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
cases = {
    "const frac": (torch.full((H, W), 0.5), torch.full((H, W), 0.25)),
    "const frac neg": (torch.full((H, W), -3.3), torch.full((H, W), -1.6)),
    "compress x": (-0.03 * xs, torch.zeros(H, W)),
    "stretch x": (0.03 * xs, torch.zeros(H, W)),
    "compress y": (torch.zeros(H, W), -0.06 * ys),
    "stretch y": (torch.zeros(H, W) + 0.3, 0.06 * ys),
    "shear": (0.05 * ys, 0.02 * xs),
}
img = torch.rand(1, 3, H, W, device=dev) * 2 - 1
for name, (fx, fy) in cases.items():
    flow = torch.stack([fx, fy])[None].to(dev).contiguous()
    a = hip.softsplat_fused(img, flow, None, "softmax", kernel="strip")
    b = hip.softsplat_fused(img, flow, None, "softmax", kernel="tile")
    d = (a - b).abs()[0].amax(0)
    bad = (d > 1e-4).nonzero()
    print("%-14s max diff %.2e, bad cells %d, first: %s" % (name, d.max().item(), bad.shape[0], bad[:6].tolist()))
