"""Feature-map splat (48 ch @288x480, softmax without metric): strip kernel (+ memset + finish) against the band kernel."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
def smooth(n, c, h, w, amp, s=32):
    lo = torch.randn(n, c, max(2, h // s), max(2, w // s), device=dev) * amp
    return torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False).contiguous()
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (C, H, W, amp) in [(48, 288, 480, 2.0), (48, 144, 240, 2.0), (48, 72, 120, 1.0)]:
    img = torch.rand(1, C, H, W, device=dev) * 2 - 1
    flow = smooth(1, 2, H, W, amp)
    a = hip.softsplat_fused(img, flow, None, "softmax", kernel="strip")
    b = hip.softsplat_fused(img, flow, None, "softmax", kernel="tile")
    print((C, H, W), "max diff %.2e" % (a - b).abs().max().item(),
          "| strip %.1f us, band %.1f us, strip->spk %.1f us" % (timeit(lambda: hip.softsplat_fused(img, flow, None, "softmax", kernel="strip")),
                                                                  timeit(lambda: hip.softsplat_fused(img, flow, None, "softmax", kernel="tile")),
                                                                  timeit(lambda: hip.softsplat_fused(img, flow, None, "softmax", want_spk=True))), flush=True)
