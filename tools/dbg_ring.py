"""Loader/consumer ring conv (variant 1) vs the barrier pipeline (variant 0) vs the register-staged split conv:
bit-exactness on the case list of dbg_spk.py, then timings per layer shape of the 4K forward."""
import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
L = hip.lib()

def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

cases = [  # (N, [src channels], up2 flags, cout, cout_store, H, W, relu, residual)
    (1, [96], [0], 96, None, 36, 60, True, False), (1, [96], [0], 96, None, 9, 15, True, True),
    (1, [48, 48], [0, 0], 48, None, 18, 30, False, False), (1, [48, 48, 4], [0, 0, 0], 96, None, 36, 60, True, False),
    (1, [48], [0], 4, None, 36, 60, False, True), (1, [48], [0], 6, 4, 9, 15, False, False),
    (2, [96], [0], 48, None, 20, 37, True, False), (1, [64, 32], [1, 0], 32, None, 24, 40, True, False),
    (1, [32, 16], [1, 0], 16, None, 48, 80, True, False), (1, [64], [0], 64, None, 36, 60, True, False),
    (1, [96], [0], 96, None, 288, 480, True, True), (1, [96], [0], 96, None, 144, 240, True, False),
    (3, [16], [0], 16, None, 5, 7, False, False), (1, [96], [0], 96, None, 288, 512, True, False),
]
bad = 0
for (N, cs, ups, cout, cst, H, W, relu, res) in cases:
    srcs = [torch.randn(N, c, H // (2 if u else 1), W // (2 if u else 1), device=dev) for c, u in zip(cs, ups)]
    wt = torch.randn(cout, sum(cs), 3, 3, device=dev) / 20
    b = torch.randn(cout, device=dev)
    rs = torch.randn(N, cst or cout, H, W, device=dev) if res else None
    up2 = [bool(u) for u in ups]
    ref = hip.conv2d(srcs, wt, b, relu=relu, residual=rs, cout_store=cst, up2=up2, precision="split")
    pk = hip.spk_pack(ref)
    for var in (0, 1, 2):
        L.fldr_debug_spk_variant(min(var, 1)); L.fldr_debug_ring_consumers(8 if var == 1 else 4)
        for prec in ("split", "fp16"):
            got, gp = hip.conv2d_spk(srcs, wt, b, relu=relu, residual=rs, cout_store=cst, up2=up2, want_f32=True, want_spk=True, precision=prec)
            torch.cuda.synchronize()
            if prec == "split":
                ok = torch.equal(ref, got) and torch.equal(pk.buf, gp.buf)
                if not ok: print("  MISMATCH variant", var, "max diff", (ref - got).abs().max().item())
                bad += not ok
            else:
                if var == 0: g16 = got
                else:
                    ok = torch.equal(g16, got); bad += not ok
                    if not ok: print("  fp16-mode MISMATCH between variants", (g16 - got).abs().max().item())
    print("N%d src%s up%s cout %d/%s %dx%d relu%d res%d checked" % (N, cs, ups, cout, cst, H, W, relu, res), flush=True)
print("MISMATCHES:", bad, " ring timeouts:", L.fldr_debug_ring_timeouts(), flush=True)

wt = torch.randn(96, 96, 3, 3, device=dev) / 30; b = torch.randn(96, device=dev)
for (cin, cout, h, w) in [(96, 96, 288, 480), (96, 96, 288, 512), (96, 96, 144, 240), (96, 96, 72, 120), (96, 96, 36, 60), (96, 96, 9, 15),
                          (96, 48, 288, 480), (48, 48, 288, 480), (48, 16, 1152, 1920), (96, 32, 576, 960), (64, 64, 288, 480), (48, 4, 288, 480)]:
    x = torch.rand(1, cin, h, w, device=dev); xp = hip.spk_pack(x); w2 = torch.randn(cout, cin, 3, 3, device=dev) / 30
    t = []
    for var in (0, 1, 2):
        L.fldr_debug_spk_variant(min(var, 1)); L.fldr_debug_ring_consumers(8 if var == 1 else 4)
        t.append(timeit(lambda: hip.conv2d_spk([xp], w2, None, relu=True, want_f32=False, want_spk=True)))
    print("%3d->%2d @%4dx%4d: barrier %.1f us, ring8 %.1f us, ring4 %.1f us" % (cin, cout, h, w, t[0], t[1], t[2]), flush=True)
print("ring timeouts:", L.fldr_debug_ring_timeouts())
