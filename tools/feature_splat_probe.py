import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.enter_test_hooks()          # variant / tuning hooks: the test build (libfldr_hip_test.so)
dev = torch.device("cuda:0"); torch.manual_seed(0); L = hip.lib()
def timeit(fn, n=20):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (h, w) in [(288, 480), (144, 240)]:
    feat = torch.rand(1, 96, h, w, device=dev) * 2 - 1
    # piecewise-constant shift (the synthetic pair) and a smooth field
    for kind in ("shift", "smooth"):
        if kind == "shift":
            up = torch.tensor([3.3, -1.2, -3.3, 1.2], device=dev).view(1, 4, 1, 1).expand(1, 4, h, w).contiguous()
        else:
            lo = (torch.rand(1, 4, 3, 5, device=dev) - 0.5) * 6
            up = F.interpolate(F.interpolate(lo, size=(h // 8, w // 8), mode="bicubic", align_corners=False), size=(h, w), mode="bilinear", align_corners=False).contiguous()
        f1, f0 = feat[:, 48:].contiguous(), feat[:, :48].contiguous()
        tp = timeit(lambda i: hip.softsplat_pair_spk(f1, up[:, :2].contiguous(), f0, up[:, 2:].contiguous(), "softmax"))
        res = []
        for v in (1, 0):
            L.fldr_debug_splat_tile_variant(v)
            res.append(timeit(lambda i: (hip.softsplat_fused(f1, up[:, :2].contiguous(), None, "softmax", kernel="tile"), hip.softsplat_fused(f0, up[:, 2:].contiguous(), None, "softmax", kernel="tile"))))
        L.fldr_debug_splat_tile_variant(1)
        print("%dx%d %s: strip pair (atomics + memset + finish, packed out) %.1f us | bands x2 (fp32 out) %.1f us | LDS-atomic tiles x2 (fp32 out) %.1f us" % (h, w, kind, tp, res[0], res[1]), flush=True)
