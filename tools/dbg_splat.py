import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd")); sys.path.insert(0, os.path.join(R, "..", "oracle"))
import fldr_hip as hip, fldr_oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for name, H, W, mk in [("zero", 8, 70, lambda f: f * 0), ("const", 8, 70, lambda f: f * 0 + 0.3), ("smooth", 8, 70, None), ("rand", 8, 70, lambda f: f)]:
    x = torch.rand(1, 1, H, W, generator=g)
    f = (torch.rand(1, 2, H, W, generator=g) - 0.5) * 6
    if mk is None:
        xs = torch.arange(W).float().view(1, 1, 1, W) * 0.03
        f = torch.cat([xs.expand(1, 1, H, W), xs.expand(1, 1, H, W) * 0.5], 1)
    else:
        f = mk(f)
    ref = O.splat_forward(x, f)
    got = hip.softsplat_fwd(x.to(dev), f.to(dev)).cpu()
    err = (got - ref).abs()
    print(name, "max err %.3e" % err.max().item(), "sum ref %.4f got %.4f" % (ref.sum().item(), got.sum().item()))
    if err.max() > 1e-4:
        ys, xs_ = torch.where(err[0, 0] > 1e-4)
        print("  bad cells (y,x,ref,got):", [(int(a), int(b), round(ref[0,0,a,b].item(),4), round(got[0,0,a,b].item(),4)) for a, b in list(zip(ys, xs_))[:12]])
