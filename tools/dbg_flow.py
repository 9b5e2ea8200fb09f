import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip, fldr_harness as Hn, pca_comp
dev = torch.device("cuda:0")
m, _, a = Hn.prepare_model(dev)
fr = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=0)).to(dev)
t = torch.tensor([[0.5]], device=dev).view(1, 1, 1, 1)
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(fr, a), a)
    flow = None
    for level in range(5, -1, -1):
        B, _, _, h, w = pyr[level].shape
        pca = pca_comp.to_pca_diff_f32(pyr[level].reshape(6, h, w), m.params[level], a, m.Mean8, m.EV8, m.meanVec8).view(1, 96, h // 8, w // 8)
        feat = m.extract_features(pca)
        if level > 0:
            flow = m.vfinet(feat, flow, t, level=level, is_training=False, normInput=pyr[level])
        else:
            # replicate flow part only
            up = hip.resize_bilinear(flow, h // 8, w // 8, mul=2.0)
            w1 = m.vfinet.softsplat(feat[:, 48:], up[:, :2]); w0 = m.vfinet.softsplat(feat[:, :48], up[:, 2:])
            f1 = m.vfinet.conv_flow1
            ca = hip.conv2d([feat[:, :48], w1], f1.weight, f1.bias); cb = hip.conv2d([feat[:, 48:], w0], f1.weight, f1.bias)
            flow = m.vfinet._chain([ca, cb, up], m.vfinet.conv_flow2, (0, 2, 4, 6, 8), final_residual=up)
        print("level", level, "flow mean", flow.mean((0, 2, 3)).tolist(), "std", flow.std((0, 2, 3)).tolist(), "absmax", flow.abs().max().item())
    lo = flow
    big = hip.resize_bilinear(0.5 * lo, 2304, 3840, mul=8.0)
    fx, fy = big[0, 2], big[0, 3]      # t*flow_01 -> flow_t0
    X = torch.arange(3840, device=dev).view(1, -1).float(); Y = torch.arange(2304, device=dev).view(-1, 1).float()
    x0 = torch.floor(X + fx); y0 = torch.floor(Y + fy)
    hal = ((x0[:, 1:] == x0[:, :-1] + 1) & (y0[:, 1:] == y0[:, :-1])).float().mean().item()
    val = ((x0[1:] == x0[:-1]) & (y0[1:] == y0[:-1] + 1)).float().mean().item()
    print("full-res flow_t0: mean", fx.mean().item(), fy.mean().item(), "std", fx.std().item(), fy.std().item())
    print("horizontal aligned fraction %.3f vertical aligned fraction %.3f" % (hal, val))
    gx = (fx[:, 1:] - fx[:, :-1]).abs(); print("mean |dfx/dx| %.4f  p99 %.3f" % (gx.mean().item(), gx.flatten()[::97].quantile(0.99).item()))
