"""Tile splat vs strip splat: agreement and timing (GPU)."""
import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def smooth(n, c, h, w, amp, s=64):
    lo = torch.randn(n, c, max(2, h // s), max(2, w // s), device=dev) * amp
    return torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False).contiguous()
for (N, C, H, W, amp, mode, has_z) in [(1, 3, 37, 150, 3.0, "softmax", True), (2, 3, 64, 200, 20.0, "softmax", True), (1, 48, 36, 60, 4.0, "softmax", False),
                                       (1, 5, 50, 70, 300.0, "average", False), (1, 3, 200, 300, 2000.0, "linear", True), (1, 7, 33, 65, 5.0, "summation", False),
                                       (1, 3, 300, 520, 0.0, "softmax", True), (1, 3, 2304, 3840, 12.0, "softmax", True), (1, 3, 2304, 3840, 40.0, "softmax", True), (1, 48, 288, 480, 3.0, "softmax", False), (1, 48, 144, 240, 3.0, "softmax", False)]:
    img = torch.rand(N, C, H, W, device=dev) * 2 - 1
    flow = smooth(N, 2, H, W, amp) if amp > 0 else torch.zeros(N, 2, H, W, device=dev)
    if amp >= 300: flow = torch.randn(N, 2, H, W, device=dev) * amp          # incoherent, wild
    z = smooth(N, 1, H, W, 1.0) if has_z else None
    a = hip.softsplat_fused(img, flow, z, mode, kernel="strip")
    b = hip.softsplat_fused(img, flow, z, mode, kernel="tile")
    torch.cuda.synchronize()
    d = (a - b).abs()
    print("N%d C%d %dx%d amp %.0f %s z%d: max diff %.2e mean %.2e (|a| max %.2f) | strip %.1f us tile %.1f us" % (
        N, C, H, W, amp, mode, has_z, d.max().item(), d.mean().item(), a.abs().max().item(),
        timeit(lambda: hip.softsplat_fused(img, flow, z, mode, kernel="strip")), timeit(lambda: hip.softsplat_fused(img, flow, z, mode, kernel="tile"))))
