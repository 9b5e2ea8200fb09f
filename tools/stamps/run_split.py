import ctypes, os, sys, torch, math
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.LIB_PATH = os.path.join(R, "libfldr_hip.so")
dev = torch.device("cuda:0")
for (h, w) in [(36, 60), (288, 480)]:
    x = torch.rand(1, 96, h, w, device=dev); wt = torch.randn(96, 96, 3, 3, device=dev) / 30; b = torch.randn(96, device=dev)
    for _ in range(5): hip.conv2d([x], wt, b, relu=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 64)()
    hip.lib().fldr_debug_read_split_stamps.argtypes = [ctypes.c_void_p]
    hip.lib().fldr_debug_read_split_stamps(buf)
    for wv in (0, 3, 4, 7):
        v = buf[wv * 8: wv * 8 + 5]; n = max(1, v[4])
        print((h, w), "wave", wv, "per chunk cycles: steps0-3 %d | store+issue %d | last step %d | barrier %d" % (v[0] / n, v[1] / n, v[2] / n, v[3] / n))
