import ctypes, os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.LIB_PATH = os.path.join(R, "libfldr_ststamp.so")
dev = torch.device("cuda:0")
torch.manual_seed(0)
def smooth(n, c, h, w, amp, s=64):
    lo = torch.randn(n, c, max(2, h // s), max(2, w // s), device=dev) * amp
    return torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False).contiguous()
for (H, W, amp, s) in [(300, 520, 0.0, 64), (2304, 3840, 3.0, 256), (2304, 3840, 12.0, 64)]:
    img = torch.rand(1, 3, H, W, device=dev) * 2 - 1
    flow = smooth(1, 2, H, W, amp, s) + 5.3 if amp > 0 else torch.zeros(1, 2, H, W, device=dev) + 0.3
    z = smooth(1, 1, H, W, 1.0)
    for _ in range(2): hip.softsplat_fused(img, flow, z, "softmax", kernel="tile")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.softsplat_fused(img, flow, z, "softmax", kernel="tile"); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 16)()
    hip.lib().fldr_debug_read_st_stamps.argtypes = [ctypes.c_void_p]
    hip.lib().fldr_debug_read_st_stamps(buf)
    v = list(buf)
    print("%dx%d amp %.0f: call %.1f us | init %d, scan+walk total %d (walk %d, of which process_block %d), flush %d, total %d cycles | blocks %d rows hit %d (fast %d, claim %d) -> %.0f cycles per block, %.0f per row"
          % (H, W, amp, e0.elapsed_time(e1) * 1e3, v[0], v[1], v[2], v[3], v[8], v[9], v[4], v[5], v[6], v[7], v[3] / max(1, v[4]), v[3] / max(1, v[5])))
