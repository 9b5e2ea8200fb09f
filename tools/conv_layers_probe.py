"""Every 3x3 convolution launch of one 4K forward: shape, time (HIP events around the call, device idle before it) and how far the launch is
from the issue bound of its 3 x fp16 matrix instructions (1.46 PF: 2.5 PF scaled to the 1.56 GHz the chip holds inside these kernels).
Usage: python tools/conv_layers_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fldr-vfi_amd"))
import torch
import fldr_hip
import fldr_harness as Hn

dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
f = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=0)).to(dev)
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(f, args), args)
t = torch.tensor([[0.5]], device=dev)
rows, on = [], [False]


def timed(name, fn, desc):
    def wrapper(*a, **k):
        if not on[0]:
            return fn(*a, **k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        rows.append((name,) + desc(*a, **k) + (e0.elapsed_time(e1) * 1e3,))
        return r
    return wrapper


def shp(x):
    return tuple(x.shape)


def d_spk(srcs, weight, bias, relu=False, residual=None, cout_store=None, up2=None, **k):
    s = [shp(x) for x in srcs]
    up = 2 if (up2 and up2[0]) else 1
    n, _, h, w = s[0]
    cout, cin = weight.shape[:2]
    return ("%s%s" % ("+".join(str(x[1]) for x in s), " up2" if up > 1 else ""), cin, cout, n, h * up, w * up, 2.0 * cin * cout * 9 * n * h * up * w * up, residual is not None)


def d_lv(srcs, weight, bias, relu=False, residuals=None, **k):
    cout, cin = weight.shape[:2]
    px = sum(x.shape[2] * x.shape[3] for x in srcs)
    return ("levels x%d" % len(srcs), cin, cout, 1, srcs[0].shape[2], srcs[0].shape[3], 2.0 * cin * cout * 9 * px, residuals is not None)


fldr_hip.conv2d_spk = timed("spk", fldr_hip.conv2d_spk, d_spk)
fldr_hip.conv2d_spk_levels = timed("levels", fldr_hip.conv2d_spk_levels, d_lv)
with torch.no_grad():
    for _ in range(3):
        Hn.interpolate(model, args, f, t, pyramid=pyr)
    on[0] = True
    Hn.interpolate(model, args, f, t, pyramid=pyr)
tot = 0.0
print("%-7s %-16s %4s %4s %2s %5s %5s %8s %8s %6s %s" % ("call", "sources", "cin", "cout", "N", "H", "W", "us", "GF alg", "eff", "res"))
for name, src, cin, cout, n, h, w, fl, res, us in rows:
    tot += us
    print("%-7s %-16s %4d %4d %2d %5d %5d %8.1f %8.2f %6.2f %s" % (name, src, cin, cout, n, h, w, us, fl / 1e9, 3 * fl / (us * 1e-6) / 1.46e15, "res" if res else ""))
print("sum %.1f us over %d launches (event-timed one by one: includes ~2-4 us of launch + event overhead each)" % (tot, len(rows)))
