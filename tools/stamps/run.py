import ctypes, os, sys, torch, math
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.LIB_PATH = os.path.join(R, "libfldr_hip.so")      # diagnostic build with in-kernel stamps
dev = torch.device("cuda:0")
H, W = 2304, 3840
cases = {
  "enc1": ([3, 3, 3, 3, 2, 2, 2, 2, 3, 3], 16, 4, 2, (H, W), None),
  "dec2": ([32, 16], 16, 3, 1, (H // 2, W // 2), [True, False]),
  "main96": ([96], 96, 3, 1, (288, 480), None),
}
for name, (parts, cout, k, stride, (Hi, Wi), up2) in cases.items():
    up2 = up2 or [False] * len(parts)
    srcs = [torch.rand(1, c, Hi // 2 if u else Hi, Wi // 2 if u else Wi, device=dev) for c, u in zip(parts, up2)]
    cin = sum(parts)
    wt = torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin * k * k); b = torch.randn(cout, device=dev)
    for _ in range(5): hip.conv2d(srcs, wt, b, stride=stride, relu=True, up2=up2)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 32)()
    hip.lib().fldr_debug_read_stamps.argtypes = [ctypes.c_void_p]
    hip.lib().fldr_debug_read_stamps(buf)
    for wv in range(2):
        v = buf[wv * 8: wv * 8 + 6]
        n = max(1, v[4])
        print(name, "wave", wv, "per chunk cycles: mfma+staging %d barrier %d | loop total %d (chunks %d)" % (v[1] / n, v[3] / n, v[5], v[4]))
