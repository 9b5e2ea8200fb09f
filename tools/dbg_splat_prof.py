import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
def smooth(n, c, h, w, amp, s=64):
    lo = torch.randn(n, c, max(2, h // s), max(2, w // s), device=dev) * amp
    return torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False).contiguous()
for (C, H, W, amp) in [(3, 37, 150, 3.0), (3, 2304, 3840, 12.0), (48, 288, 480, 3.0)]:
    img = torch.rand(1, C, H, W, device=dev) * 2 - 1
    flow = smooth(1, 2, H, W, amp); z = smooth(1, 1, H, W, 1.0) if C == 3 else None
    for _ in range(3):
        hip.softsplat_fused(img, flow, z, "softmax", kernel="tile")
    torch.cuda.synchronize()
