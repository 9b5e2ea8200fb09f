"""Split-packed persistent conv vs the register-staged split conv: bit-exactness and timing (GPU)."""
import os, sys, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)

def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

cases = [  # (N, [src channels], up2 flags, cout, cout_store, H, W, relu, residual)
    (1, [96], [0], 96, None, 36, 60, True, False),
    (1, [96], [0], 96, None, 9, 15, True, True),
    (1, [48, 48], [0, 0], 48, None, 18, 30, False, False),
    (1, [48, 48, 4], [0, 0, 0], 96, None, 36, 60, True, False),
    (1, [48], [0], 4, None, 36, 60, False, True),
    (1, [48], [0], 6, 4, 9, 15, False, False),
    (2, [96], [0], 48, None, 20, 37, True, False),
    (1, [64, 32], [1, 0], 32, None, 24, 40, True, False),
    (1, [32, 16], [1, 0], 16, None, 48, 80, True, False),
    (1, [64], [0], 64, None, 36, 60, True, False),
    (1, [96], [0], 96, None, 288, 480, True, True),
    (1, [96], [0], 96, None, 144, 240, True, False),
]
bad = 0
for (N, cs, ups, cout, cst, H, W, relu, res) in cases:
    srcs = [torch.randn(N, c, H // (2 if u else 1), W // (2 if u else 1), device=dev) for c, u in zip(cs, ups)]
    wt = torch.randn(cout, sum(cs), 3, 3, device=dev) / 20
    b = torch.randn(cout, device=dev)
    rs = torch.randn(N, cst or cout, H, W, device=dev) if res else None
    ref = hip.conv2d(srcs, wt, b, relu=relu, residual=rs, cout_store=cst, up2=[bool(u) for u in ups], precision="split")
    got, gp = hip.conv2d_spk(srcs, wt, b, relu=relu, residual=rs, cout_store=cst, up2=[bool(u) for u in ups], want_f32=True, want_spk=True)
    torch.cuda.synchronize()
    same = torch.equal(ref, got)
    pk = hip.spk_pack(ref)
    samep = torch.equal(pk.buf, gp.buf)
    md = (ref - got).abs().max().item()
    print("N%d src%s up%s cout %d/%s %dx%d relu%d res%d: fp32 bit-exact %s (max diff %.2e), packed bit-exact %s" %
          (N, cs, ups, cout, cst, H, W, relu, res, same, md, samep))
    bad += (not same) + (not samep)
    # packed -> packed chain equals fp32 chain
    if len(cs) == 1 and not ups[0] and cst is None and cout % 8 == 0:
        w2 = torch.randn(48, cout, 3, 3, device=dev) / 20
        r2 = hip.conv2d([ref], w2, None, precision="split")
        g2 = hip.conv2d_spk([gp], w2, None)
        ok2 = torch.equal(r2, g2)
        print("    chained through the packed tensor: bit-exact %s" % ok2)
        bad += not ok2
print("MISMATCHES:", bad)

x = torch.rand(1, 96, 288, 480, device=dev); wt = torch.randn(96, 96, 3, 3, device=dev) / 30; b = torch.randn(96, device=dev)
xp = hip.spk_pack(x)
print("old split 96->96 @288x480: %.1f us" % timeit(lambda: hip.conv2d([x], wt, b, relu=True, precision="split")))
for wg in (32,):
    hip.lib().fldr_debug_spk_wgs_per_xcd(wg)
    print("spk wgs/xcd %d: f32 out %.1f us | packed out %.1f us | pack kernel alone %.1f us" % (
        wg, timeit(lambda: hip.conv2d_spk([xp], wt, b, relu=True)),
        timeit(lambda: hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True)),
        timeit(lambda: hip.spk_pack(x))))
for (h, w) in [(144, 240), (72, 120), (36, 60), (9, 15)]:
    x = torch.rand(1, 96, h, w, device=dev); xp = hip.spk_pack(x)
    print("%dx%d: old %.1f us, spk %.1f us" % (h, w, timeit(lambda: hip.conv2d([x], wt, b, relu=True, precision="split")),
                                              timeit(lambda: hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True))))
for (cin, cout, h, w) in [(96, 48, 288, 480), (48, 48, 288, 480), (48, 16, 1152, 1920), (96, 32, 576, 960), (64, 64, 288, 480)]:
    x = torch.rand(1, cin, h, w, device=dev); xp = hip.spk_pack(x); w2 = torch.randn(cout, cin, 3, 3, device=dev) / 30
    print("%d->%d @%dx%d: old %.1f us, spk %.1f us" % (cin, cout, h, w, timeit(lambda: hip.conv2d([x], w2, None, relu=True, precision="split")),
                                                      timeit(lambda: hip.conv2d_spk([xp], w2, None, relu=True, want_f32=False, want_spk=True))))
