"""fp64-LDS-atomic tile splat (fldr_softsplat_acc64) against the kernels it replaces, at the shapes of a 4K forward:
the two level-0 image splats (bounds from the low-resolution flow) and the feature-splat pair of the finest levels, on a
piecewise-constant shift (the bench's synthetic pairs) and on a smoothly varying field."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.enter_test_hooks()          # variant hooks: the test build (libfldr_hip_test.so)
dev = torch.device("cuda:0"); torch.manual_seed(0); L = hip.lib()


def timeit(fn, n=20):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def flows(kind, h, w):
    if kind == "shift":
        return torch.tensor([3.3, -1.2, -3.3, 1.2], device=dev).view(1, 4, 1, 1).expand(1, 4, h, w).contiguous()
    lo = (torch.rand(1, 4, 3, 5, device=dev) - 0.5) * 6
    return F.interpolate(lo, size=(h, w), mode="bicubic", align_corners=False).contiguous()


def images():
    H, W, up = 2304, 3840, 8
    frames = [torch.rand(1, 3, 2, H, W, device=dev) * 2 - 1 for _ in range(3)]
    t4 = torch.full((1, 1, 1, 1), 0.5, device=dev)
    for kind in ("shift", "smooth"):
        lo = flows(kind, H // up, W // up)
        ft0 = (F.interpolate(t4 * lo[:, 2:], scale_factor=up, mode="bilinear", align_corners=False) * up).contiguous()
        ft1 = (F.interpolate((1 - t4) * lo[:, :2], scale_factor=up, mode="bilinear", align_corners=False) * up).contiguous()
        z0 = -torch.rand(1, 1, H, W, device=dev) * 3; z1 = -torch.rand(1, 1, H, W, device=dev) * 3
        b0 = hip.splat_bounds_upsampled(lo[:, 2:], t4, 1, up, H, W); b1 = hip.splat_bounds_upsampled(lo[:, :2], t4, 2, up, H, W)
        bw = hip.splat_bounds_upsampled_pair(lo, t4, "images", up, H, W)
        def band(i):
            f = frames[i % 3]
            hip.softsplat_fused(f[:, :, 0], ft0, z0, "softmax", kernel="tile", bounds_ws=b0)
            hip.softsplat_fused(f[:, :, 1], ft1, z1, "softmax", kernel="tile", bounds_ws=b1)
        def acc(i):
            f = frames[i % 3]
            hip.softsplat_acc64([f[:, :, 0], f[:, :, 1]], [ft0, ft1], [z0, z1], "softmax", bounds_ws=bw)
        tb, ta = timeit(band), timeit(acc)
        f = frames[0]
        r = hip.softsplat_acc64([f[:, :, 0], f[:, :, 1]], [ft0, ft1], [z0, z1], "softmax", bounds_ws=[b0, b1])
        q = hip.softsplat_fused(f[:, :, 0], ft0, z0, "softmax", kernel="tile", bounds_ws=b0)
        print("images 2304x3840 %s: band x2 %.1f us | acc64 pair %.1f us | max |acc64 - band| %.2e" % (kind, tb, ta, (r[0] - q).abs().max().item()), flush=True)


def features():
    for (h, w) in [(288, 480), (144, 240), (72, 120), (36, 60), (18, 30)]:
        feat = torch.rand(1, 96, h, w, device=dev) * 2 - 1
        for kind in ("shift", "smooth"):
            prev = flows(kind, h // 2, w // 2)
            up = hip.resize_bilinear(prev, h, w, mul=2.0)
            f1, f0 = feat[:, 48:], feat[:, :48]
            f1c, f0c, ua, ub = f1.contiguous(), f0.contiguous(), up[:, :2].contiguous(), up[:, 2:].contiguous()
            tp = timeit(lambda i: hip.softsplat_pair_spk(f1c, ua, f0c, ub, "softmax"))
            ta = timeit(lambda i: hip.softsplat_acc64([f1, f0], [up[:, :2], up[:, 2:]], None, "softmax", want_f32=False, want_spk=True, spk_batch=True,
                                                      bounds_ws=hip.splat_bounds_upsampled_pair(prev, None, "features", 2.0, h, w) if h * w > 2304 else None))
            tf = None
            if hasattr(hip.lib(), "fldr_debug_splat_group_fold"):
                hip.lib().fldr_debug_splat_group_fold(1)
                tf = timeit(lambda i: hip.softsplat_acc64([f1, f0], [up[:, :2], up[:, 2:]], None, "softmax", want_f32=False, want_spk=True, spk_batch=True,
                                                          bounds_ws=hip.splat_bounds_upsampled_pair(prev, None, "features", 2.0, h, w) if h * w > 2304 else None))
                hip.lib().fldr_debug_splat_group_fold(0)
            a = hip.softsplat_acc64([f1, f0], [up[:, :2], up[:, 2:]], None, "softmax", want_f32=True, want_spk=False)
            s = hip.softsplat_fused(f1c, ua, None, "softmax", kernel="strip")
            print("features %dx%d %s: strip pair %.1f us | acc64 pair %.1f us%s | max |acc64 - strip| %.2e"
                  % (h, w, kind, tp, ta, "" if tf is None else " (all channel groups in one workgroup: %.1f us)" % tf, (a[0] - s).abs().max().item()), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "features"): features()
    if which in ("all", "images"): images()
