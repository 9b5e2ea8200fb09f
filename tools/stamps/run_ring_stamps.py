import ctypes, os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.LIB_PATH = os.path.join(R, "libfldr_rstamp.so")
dev = torch.device("cuda:0")
# usage: run_ring_stamps.py [cin cout h w]   (default: the 96 -> 96 layer at three sizes)
CIN, COUT = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 4 else (96, 96)
SHAPES = [(int(sys.argv[3]), int(sys.argv[4]), 0)] if len(sys.argv) > 4 else [(36, 60, 0), (272, 480, 0), (272, 480, 1)]
wt = torch.randn(COUT, CIN, 3, 3, device=dev) / 30; b = torch.randn(COUT, device=dev)
for (h, w, cold) in SHAPES:
    xs = [hip.spk_pack(torch.rand(1, CIN, h, w, device=dev)) for _ in range(6 if cold else 1)]     # cold: 6 rotating inputs (600 MB > Infinity Cache)
    for i in range(12): hip.conv2d_spk([xs[i % len(xs)]], wt, b, relu=True, want_f32=False, want_spk=True)
    torch.cuda.synchronize()
    xp = xs[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 32)()
    hip.lib().fldr_debug_read_ring_stamps.argtypes = [ctypes.c_void_p]
    hip.lib().fldr_debug_read_ring_stamps(buf)
    print((h, w), "cold" if cold else "hot", "launch %.1f us" % (e0.elapsed_time(e1) * 1e3))
    for k, name in enumerate(("wg0 consumer0", "wg0 loader4", "wg101 consumer0", "wg101 loader4")):
        v = buf[k * 8: k * 8 + 8]; n = max(1, v[5])
        if k % 2 == 0:
            print("  %-16s iters %3d | per iteration cycles: wait-FULL %5d steps %5d finish/store %5d (of it: unit decode %d, fast-path finish %d, its first block %d) | loop total %d cycles" % (name, v[5], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[7] / n, v[6]))
        else:
            print("  %-16s fills %3d | per fill cycles: prepare %5d wait-FREE %5d fire %5d wait-landed %5d | loop total %d cycles" % (name, v[5], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[6]))

    tr = (ctypes.c_uint64 * (4 * 24 * 4))()
    hip.lib().fldr_debug_read_ring_trace.argtypes = [ctypes.c_void_p]
    hip.lib().fldr_debug_read_ring_trace(tr)
    t0 = min(v for v in tr if v)
    names = ("consumer0", "consumer4", "loader0", "loader3")
    ev = (("poll", "full", "steps", "fin"), ("poll", "full", "steps", "fin"), ("wantfree", "free", "fired", "signalled"), ("wantfree", "free", "fired", "signalled"))
    for it in range(min(14, int(buf[5]))):
        print("  it %2d " % it + " | ".join("%s %s" % (names[w], " ".join("%6d" % (tr[(w * 24 + it) * 4 + e] - t0) for e in range(4))) for w in range(4)))
