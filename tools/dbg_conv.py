import sys, os, math, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
for (N, H, W) in [(1, 22, 70), (1, 24, 70), (1, 22, 72), (1, 32, 64), (1,16,34)]:
    src = torch.rand(N, 16, H // 2, W // 2, generator=g) * 2 - 1
    wt = torch.randn(6, 16, 3, 3, generator=g) / 12
    ref = F.conv2d(F.interpolate(src, scale_factor=2, mode="nearest"), wt, None, padding=1)
    got = hip.conv2d([src.to(dev)], wt.to(dev), None, up2=[True]).cpu()
    err = (got - ref).abs()
    bad = (err > 1e-4)
    ys, xs = torch.where(bad.any(1)[0])
    print((N, H, W), "max err %.3e" % err.max().item(), "bad rows", sorted(set(ys.tolist()))[:40], "bad cols", sorted(set(xs.tolist()))[:80])
