#!/usr/bin/env python3
"""Per-kernel timings at the shapes of a 4K forward (GPU box; FLDR_LIB=<path> selects an experimental library built with
tools/build_lib_variant.sh or tools/stamps/build_variant.sh).  One script instead of the per-kernel dbg_*.py of round 1.

    python tools/kernel_bench.py conv      # fldr_conv2d_spk: barrier pipeline / ring with 8 / 4 consumer waves, per layer shape
    python tools/kernel_bench.py conv_cold # the dominant conv with cache-resident vs rotating inputs
    python tools/kernel_bench.py conv_tw   # ring pipeline on 8x32 vs 8x16 tiles
    python tools/kernel_bench.py s2        # stride-2 encoders: tile-grid shift 0 / 15 / 31 of the persistent kernel
    python tools/kernel_bench.py dec3      # dec3 + softmax/T + blend: tile-grid shift 0 / 16
    python tools/kernel_bench.py pca       # PCA of a whole pyramid: per-level one-pass kernels vs the two pyramid launches
    python tools/kernel_bench.py band      # image band splat (bounds + band)
    python tools/kernel_bench.py gather    # feature splats of one level: deterministic gather vs atomic scatter + finish
"""
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


def conv():
    for (cin, cout, h, w) in [(96, 96, 288, 480), (96, 96, 288, 512), (96, 96, 144, 240), (96, 96, 72, 120), (96, 96, 9, 15), (96, 48, 288, 480),
                              (48, 48, 288, 480), (48, 16, 1152, 1920), (96, 32, 576, 960), (64, 64, 288, 480), (48, 4, 288, 480)]:
        x = torch.rand(1, cin, h, w, device=dev); xp = hip.spk_pack(x); w2 = torch.randn(cout, cin, 3, 3, device=dev) / 30
        t = []
        L.fldr_debug_ring32(0)                               # the 16x16x32 kernels in the first three columns
        for var in (0, 1, 2):
            L.fldr_debug_spk_variant(min(var, 1)); L.fldr_debug_ring_consumers(8 if var == 1 else 4)
            t.append(timeit(lambda i: hip.conv2d_spk([xp], w2, None, relu=True, want_f32=False, want_spk=True), 40))
        extra = ""
        L.fldr_debug_spk_variant(1); L.fldr_debug_ring_consumers(8)
        if cout in (64, 96):                                 # the 32x32x16 ring kernel
            L.fldr_debug_ring32(2)
            extra += ", ring32 %.1f us" % timeit(lambda i: hip.conv2d_spk([xp], w2, None, relu=True, want_f32=False, want_spk=True), 40)
        if cout <= 16 and cin <= 64:                         # the resident-weight ring applies (test build, off by default): beside the default
            L.fldr_debug_ring_resident(1)
            for cons in (8, 4):
                L.fldr_debug_ring_consumers(cons)
                extra += ", resident weights ring%d %.1f us" % (cons, timeit(lambda i: hip.conv2d_spk([xp], w2, None, relu=True, want_f32=False, want_spk=True), 40))
            L.fldr_debug_ring_resident(0)
        print("%3d->%2d @%4dx%4d: barrier %.1f us, ring8 %.1f us, ring4 %.1f us%s" % (cin, cout, h, w, t[0], t[1], t[2], extra), flush=True)
    L.fldr_debug_spk_variant(1); L.fldr_debug_ring_consumers(8); L.fldr_debug_ring32(1)
    print("ring timeouts:", L.fldr_debug_ring_timeouts())


def conv_cold():
    """The 96->96 convolution at the level-0 geometries of the bench (272x480) and of X-Test (288x512): the same input every
    launch (Infinity-Cache resident) against 6 rotating inputs / outputs (600 MB: every launch streams from HBM)."""
    for (h, w) in [(272, 480), (288, 480), (288, 512)]:
        xs = [hip.spk_pack(torch.rand(1, 96, h, w, device=dev)) for _ in range(6)]
        w2 = torch.randn(96, 96, 3, 3, device=dev) / 30
        hot = timeit(lambda i: hip.conv2d_spk([xs[0]], w2, None, relu=True, want_f32=False, want_spk=True), 40)
        cold = timeit(lambda i: hip.conv2d_spk([xs[i % 6]], w2, None, relu=True, want_f32=False, want_spk=True), 42)
        tiles = -(-h // 8) * -(-w // 32)
        print("96->96 @%dx%d (%d units, %.2f rounds of 256): same input %.1f us, rotating inputs %.1f us" % (h, w, 2 * tiles, 2 * tiles / 256, hot, cold), flush=True)


def conv_tw():
    """8x32 against 8x16 tiles of the ring pipeline at the second pyramid level of a 4K pair (144x240: 288 wide units on 256
    persistent workgroups) and at level 0."""
    for (n, cin, cout, h, w) in [(1, 96, 96, 144, 240), (2, 96, 48, 144, 240), (1, 96, 48, 144, 240), (1, 48, 48, 144, 240), (1, 96, 96, 288, 480),
                                 (1, 96, 96, 72, 120), (1, 96, 32, 576, 960), (1, 48, 16, 1152, 1920), (1, 96, 96, 144, 256), (1, 96, 96, 288, 512)]:
        xs = [hip.spk_pack(torch.rand(n, cin, h, w, device=dev)) for _ in range(3)]
        w2 = torch.randn(cout, cin, 3, 3, device=dev) / 30
        row = []
        for tw in (32, 16, 0):
            L.fldr_debug_ring_tile_width(tw)
            row.append("tw %2d: %.1f us" % (tw, timeit(lambda i: hip.conv2d_spk([xs[i % 3]], w2, None, relu=True, want_f32=False, want_spk=True), 30)))
        print("N%d %3d->%2d @%4dx%4d  " % (n, cin, cout, h, w) + " | ".join(row), flush=True)
    L.fldr_debug_ring_tile_width(0)


def s2():
    for (cin, cout, h, w) in [(26, 16, 2304, 3840), (16, 32, 1152, 1920), (26, 16, 2304, 4096)]:
        xs = [torch.rand(1, cin, h, w, device=dev) * 2 - 1 for _ in range(3)]      # rotated: 3 x 0.9 GB > Infinity Cache
        wt = torch.randn(cout, cin, 4, 4, device=dev) / 20; b = torch.randn(cout, device=dev)
        row = []
        for sh, v4 in ((0, 1), (15, 0), (15, 1), (31, 1)):
            L.fldr_debug_s2_xshift(sh); L.fldr_debug_s2_vec4(v4)
            row.append("shift %2d%s: %.1f us" % (sh, " 16-B loads" if v4 and sh & 1 else "", timeit(lambda i: hip.conv2d([xs[i % 3]], wt, b, stride=2, relu=True, precision="split", want_f32=True, want_spk=True))))
        print("%d->%d @%dx%d  " % (cin, cout, h, w) + " | ".join(row), flush=True)
    L.fldr_debug_s2_xshift(-1); L.fldr_debug_s2_vec4(1)


def s2spk():
    """enc2 / enc3 on a split-packed source (fldr_conv2d_s2_spk; enc3 as its two 32-channel halves in one launch), rotating inputs."""
    for (cin, cout, h, w) in [(16, 32, 1152, 1920), (32, 64, 576, 960)]:
        xs = [hip.spk_pack(torch.rand(1, cin, h, w, device=dev) * 2 - 1) for _ in range(3)]
        wt = torch.randn(cout, cin, 4, 4, device=dev) / 20; b = torch.randn(cout, device=dev)
        if cout == 64:
            halves = [(wt[k:k + 32].contiguous(), b[k:k + 32].contiguous()) for k in (0, 32)]
            t = timeit(lambda i: hip.conv2d_s2_spk_pair(xs[i % 3], halves, relu=True), 30)
        else:
            t = timeit(lambda i: hip.conv2d_s2_spk(xs[i % 3], wt, b, relu=True, want_f32=False, want_spk=True), 30)
        print("%d->%d @%dx%d packed source: %.1f us" % (cin, cout, h, w, t), flush=True)


def dec3():
    H, W = 2304, 3840
    sets = [(torch.rand(1, 16, H // 2, W // 2, device=dev), [torch.rand(1, 3, H, W, device=dev) * 2 - 1 for _ in range(6)]) for _ in range(2)]
    wt = torch.randn(6, 16, 3, 3, device=dev) / 6; bs = torch.randn(6, device=dev) * 0.3; t = torch.tensor([[0.5]], device=dev)
    for xs in (0, 16, 0, 16):
        L.fldr_debug_dec3_xshift(xs)
        print("x shift %2d: %.1f us" % (xs, timeit(lambda i: hip.dec3_synth(sets[i % 2][0], wt, bs, sets[i % 2][1], t, 1.5616), 16)), flush=True)
    L.fldr_debug_dec3_xshift(-1)
    psets = [hip.spk_pack(sets[i][0]) for i in range(2)]
    print("split-packed source, matrix cores: %.1f us" % timeit(lambda i: hip.dec3_synth(psets[i % 2], wt, bs, sets[i % 2][1], t, 1.5616), 16), flush=True)
    print("fp32 source, vector ALUs         : %.1f us" % timeit(lambda i: hip.dec3_synth(sets[i % 2][0], wt, bs, sets[i % 2][1], t, 1.5616), 16), flush=True)
    outs = {}
    for xc in (0, 1, 0, 1):
        L.fldr_debug_dec3_xcd(xc)
        print("tile order %s: %.1f us" % ("XCD-contiguous" if xc else "row-major     ", timeit(lambda i: hip.dec3_synth(sets[i % 2][0], wt, bs, sets[i % 2][1], t, 1.5616), 16)), flush=True)
        outs[xc] = hip.dec3_synth(sets[0][0], wt, bs, sets[0][1], t, 1.5616)
    print("identical:", bool(torch.equal(outs[0], outs[1])))
    L.fldr_debug_dec3_xcd(1)


def pca():
    import fldr_harness as Hn
    model, _, args = Hn.prepare_model(dev)
    ev, mean, mv = model.EV8.detach(), model.Mean8.detach(), model.meanVec8.detach()
    pyrs = [[(torch.rand(6, 2304 >> l, 3840 >> l, device=dev) * 2 - 1) for l in range(6)] for _ in range(3)]
    print("pyramid, rotating inputs: per-level %.1f us, pyramid kernels %.1f us" % (
        timeit(lambda i: [hip.pca_project_stream(p, ev, mean, mv, want_spk=True) for p in pyrs[i % 3]]),
        timeit(lambda i: hip.pca_project_pyramid(pyrs[i % 3], ev, mean, mv, want_f32=True, want_spk=True))))
    for rm, what in ((1 << 40, "recompute in pass B          "), (4 << 20, "projections parked >= 4 MB   "), (0, "projections parked, all levels")):
        print("pyramid, %s: %.1f us" % (what, timeit(lambda i: hip.pca_project_pyramid(pyrs[i % 3], ev, mean, mv, want_f32=True, want_spk=True, raw_min_bytes=rm))), flush=True)
    print("level 0 only            : per-level %.1f us, pyramid kernels %.1f us" % (
        timeit(lambda i: hip.pca_project_stream(pyrs[i % 3][0], ev, mean, mv, want_spk=True)),
        timeit(lambda i: hip.pca_project_pyramid(pyrs[i % 3][:1], ev, mean, mv, want_f32=True, want_spk=True))))


def band():
    H, W = 2304, 3840
    sets = []
    for k in range(3):
        lo = (torch.rand(1, 2, H // 8, W // 8, device=dev) - 0.5) * 4 + torch.tensor([6.0, 4.0], device=dev).view(1, 2, 1, 1)
        flow = hip.resize_bilinear(hip.resize_bilinear(lo, H // 2, W // 2), H, W)
        sets.append((torch.rand(1, 3, H, W, device=dev) * 2 - 1, flow, torch.rand(1, 1, H, W, device=dev) * -2))
    print("band splat (bounds + band): %.1f us" % timeit(lambda i: hip.softsplat_fused(*sets[i % 3], "softmax", kernel="tile"), 16), flush=True)


def gather():
    for (h, w) in [(288, 480), (144, 240), (72, 120), (18, 30)]:
        feat = torch.rand(1, 96, h, w, device=dev) * 2 - 1
        for amp in (1.0, 4.0):
            lo = (torch.rand(1, 4, max(h // 8, 2), max(w // 8, 2), device=dev) - 0.5) * amp + torch.tensor([amp, -amp / 2, -amp, amp / 3], device=dev).view(1, 4, 1, 1)
            up = F.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)
            f1, f0 = feat[:, 48:], feat[:, :48]
            tg = timeit(lambda i: hip.softsplat_gather([f1, f0], [up[:, :2], up[:, 2:]], None, "softmax"))
            ts = timeit(lambda i: (hip.softsplat_fused(f1, up[:, :2], None, "softmax", want_spk=True), hip.softsplat_fused(f0, up[:, 2:], None, "softmax", want_spk=True)))
            print("%3dx%3d flow ~%4.1f px: gather %.1f us, scatter+finish %.1f us" % (h, w, amp, tg, ts), flush=True)


def corr():
    """PWC cost volume at the decoder shapes of a 4K pair (N = 2: both directions): synchronous staging vs the LDS-DMA double buffer
    with 8- / 16-channel chunks; algorithmic bytes (2 C planes in, 81 out) over the launch time against 8 TB/s."""
    for (n, c, h, w) in [(2, 196, 34, 60), (2, 128, 68, 120), (2, 96, 136, 240), (2, 64, 272, 480), (2, 32, 544, 960)]:
        a = [torch.randn(n, c, h, w, device=dev) for _ in range(3)]; b = [torch.randn(n, c, h, w, device=dev) for _ in range(3)]
        t = []
        for variant, cc in ((0, 8), (1, 8), (1, 16)):
            L.fldr_debug_corr_variant(variant); L.fldr_debug_corr_chunk(cc)
            t.append(timeit(lambda i: hip.correlation_fwd(a[i % 3], b[i % 3]), 30))
        mb = n * h * w * (2 * c + 81) * 4 / 1e6
        L.fldr_debug_corr_variant(1); L.fldr_debug_corr_chunk(8)
        L.fldr_debug_corr_xcd(0)
        trm = timeit(lambda i: hip.correlation_fwd(a[i % 3], b[i % 3]), 30)
        o0 = hip.correlation_fwd(a[0], b[0])
        L.fldr_debug_corr_xcd(1)
        same = bool(torch.equal(o0, hip.correlation_fwd(a[0], b[0])))
        print("%3d ch @%4dx%4d (%6.1f MB): sync %.1f us, dma8 %.1f us (%.2f of 8 TB/s), dma16 %.1f us; dma8 row-major tile order %.1f us (same bits: %s)" % (
            c, h, w, mb, t[0], t[1], mb / t[1] / 8.0, t[2], trm, same), flush=True)
    L.fldr_debug_corr_variant(1); L.fldr_debug_corr_chunk(8)


def prep():
    """fldr_level0_prep at the 4K shape (x8 upsampling of a 288x480 flow pair, three rotated frame pairs)."""
    H, W, up = 2304, 3840, 8
    noise = float(os.environ.get("PREP_NOISE", "0.02"))                   # low-resolution pixels: 0.02 = the bench's rigid shift, 1.5 = incoherent gathers
    lo = torch.tensor([-0.75, -0.5, 0.75, 0.5], device=dev).view(1, 4, 1, 1) + torch.randn(1, 4, H // up, W // up, device=dev) * noise
    frames = [torch.rand(1, 3, 2, H, W, device=dev) * 2 - 1 for _ in range(3)]
    t4 = torch.tensor([0.5], device=dev).view(1, 1, 1, 1)
    for rep in range(2):
        us = timeit(lambda i: hip.level0_prep(lo, frames[i % 3][:, :, 0], frames[i % 3][:, :, 1], t4, H, W, 20.0, 20.0, withmask=True, want_z=True), 16)
        print("level0_prep 2304x3840 (flow noise %g): %.1f us" % (noise, us), flush=True)
    sts = [hip.level0_prep(lo, frames[k][:, :, 0], frames[k][:, :, 1], t4, H, W, 20.0, 20.0, withmask=True, want_z=True, phase=1) for k in range(3)]
    p1 = timeit(lambda i: hip.level0_prep(lo, frames[i % 3][:, :, 0], frames[i % 3][:, :, 1], t4, H, W, 20.0, 20.0, withmask=True, want_z=True, phase=1), 16)
    p2 = timeit(lambda i: hip.level0_prep(None, None, None, None, H, W, 0, 0, state=sts[i % 3]), 16)
    nz = timeit(lambda i: hip.level0_prep(lo, frames[i % 3][:, :, 0], frames[i % 3][:, :, 1], t4, H, W, 20.0, 20.0, withmask=True, want_z=False), 16)
    print("  phase 1 (z + flow_t) %.1f us, phase 2 (flowback + im_tot) %.1f us, everything without z %.1f us" % (p1, p2, nz), flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["conv", "s2", "s2spk", "dec3", "pca", "band", "gather"]
    for name in which:
        print("----", name, "(%s)" % os.environ.get("FLDR_LIB", "product library").split("/")[-1], flush=True)
        globals()[name]()
