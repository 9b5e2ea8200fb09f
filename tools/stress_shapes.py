"""Odd-shape stress of the kernels with variant hooks / unfused counterparts: persistent vs per-tile stride-2 conv (bit
identity), band splat vs strip splat (summation order only), fused level-0 prep vs the unfused kernels (bit identity)."""
import os, sys, random, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
hip.enter_test_hooks()          # variant / tuning hooks: the test build (libfldr_hip_test.so)
dev = torch.device("cuda:0")
random.seed(0); torch.manual_seed(0)
bad = 0
for it in range(40):
    N = random.choice([1, 1, 2, 3]); cin = random.choice([1, 3, 4, 5, 16, 26, 31, 33, 64]); cout = random.choice([1, 7, 16, 17, 32])
    H = random.choice([4, 6, 10, 18, 34, 66, 130]); W = random.choice([4, 6, 12, 64, 66, 68, 130, 134, 258])
    parts, left = [], cin
    while left > 0:
        c = min(left, random.choice([1, 2, 3, 5, 16])); parts.append(c); left -= c
    parts = parts[:10] if len(parts) <= 10 else [cin]
    srcs = [torch.randn(N, c, H, W, device=dev) for c in parts]
    wt = torch.randn(cout, cin, 4, 4, device=dev) / 8; b = torch.randn(cout, device=dev)
    outs = []
    for mode, shift, v4 in ((0, -1, 1), (1, -1, 1), (1, 15, 1), (1, 15, 0), (1, random.choice([1, 3, 7, 31]), 1)):   # per-tile; persistent: auto, 16-byte / 4-byte staging, odd shifts
        hip.lib().fldr_debug_s2_persistent(mode); hip.lib().fldr_debug_s2_xshift(shift); hip.lib().fldr_debug_s2_vec4(v4)
        o, sp = hip.conv2d(srcs, wt, b, stride=2, relu=bool(it & 1), precision="split", want_spk=True)
        outs.append((o.clone(), sp.buf.clone()))
    ok = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    bad += not ok
    if not ok: print("s2 MISMATCH", N, parts, cout, H, W)
hip.lib().fldr_debug_s2_persistent(1); hip.lib().fldr_debug_s2_xshift(-1); hip.lib().fldr_debug_s2_vec4(1)
print("stride-2 persistent vs per-tile: 40 shapes,", bad, "mismatches", flush=True)
bad = 0
for it in range(30):          # packed-source stride-2 encoders: LDS-DMA kernel (17..32 outputs) vs the register-staged kernel (fp32 rounding apart)
    N = random.choice([1, 1, 2, 3]); cin = random.choice([8, 16, 24, 32, 40]); cout = random.choice([17, 20, 24, 31, 32])
    H = random.choice([4, 6, 10, 18, 34, 66, 130]); W = random.choice([4, 6, 12, 64, 66, 68, 130, 134, 258])
    x = torch.relu(torch.randn(N, cin, H, W, device=dev)) * 3
    xp = hip.spk_pack(x)
    wt = torch.randn(cout, cin, 4, 4, device=dev) / (cin * 16) ** 0.5; b = torch.randn(cout, device=dev)
    if not hip.s2_spk_ok(wt): continue
    outs = []
    for dma in (0, 1):
        hip.lib().fldr_debug_s2_dma(dma)
        o, sp = hip.conv2d_s2_spk(xp, wt, b, relu=bool(it & 1), want_f32=True, want_spk=True)
        outs.append((o.clone(), sp.float()))
    hip.lib().fldr_debug_s2_dma(1)
    tol = 2e-6 * float(outs[0][0].abs().max()) + 1e-7
    ok = (outs[0][0] - outs[1][0]).abs().max().item() <= tol and (outs[1][1] - outs[1][0]).abs().max().item() <= tol
    bad += not ok
    if not ok: print("s2 dma MISMATCH", N, cin, cout, H, W, (outs[0][0] - outs[1][0]).abs().max().item(), tol)
print("stride-2 packed source, LDS-DMA vs register-staged: 30 shapes,", bad, "mismatches", flush=True)
bad = 0
for it in range(30):
    N = random.choice([1, 2]); C = random.choice([1, 2, 3]); H = random.choice([1, 3, 11, 12, 13, 25, 70]); W = random.choice([1, 2, 55, 56, 57, 113, 300])
    mode = random.choice(["summation", "average", "linear", "softmax"])
    x = torch.rand(N, C, H, W, device=dev) * 2 - 1
    flow = (torch.rand(N, 2, H, W, device=dev) - 0.5) * random.choice([0.0, 1.0, 8.0, 100.0])
    if random.random() < 0.5:
        flow = torch.nn.functional.interpolate(torch.randn(N, 2, 2, 3, device=dev) * 5, size=(H, W), mode="bilinear").contiguous()
    z = torch.rand(N, 1, H, W, device=dev) + 0.1 if mode in ("linear", "softmax") else None
    a = hip.softsplat_fused(x, flow, z, mode, kernel="tile"); b2 = hip.softsplat_fused(x, flow, z, mode, kernel="strip")
    err = (a - b2).abs().max().item()
    if not err < 5e-5: bad += 1; print("splat MISMATCH", N, C, H, W, mode, err)
print("band vs strip splat: 30 shapes,", bad, "mismatches", flush=True)
bad = 0
for it in range(20):
    N = random.choice([1, 2]); h = random.choice([1, 2, 3, 9, 20]); w = random.choice([1, 2, 5, 15, 33]); up = random.choice([2, 3, 4, 5, 8])
    H, W = h * up, w * up
    if W < 2: continue
    lo = (torch.rand(N, 4, h, w, device=dev) - 0.5) * random.choice([1.0, 6.0, 50.0])
    x = torch.rand(N, 3, 2, H, W, device=dev) * 2 - 1
    I0, I1 = x[:, :, 0], x[:, :, 1]
    t4 = torch.rand(N, 1, 1, 1, device=dev)
    r = hip.level0_prep(lo, I0, I1, t4, H, W, -1.9, -1.8, withmask=bool(it & 1), want_z=True)
    both = hip.resize_bilinear(lo, H, W, mul=float(up)); f10, f01 = both[:, :2], both[:, 2:]
    I0c, I1c = I0.contiguous(), I1.contiguous()
    ok = torch.equal(r["z0"], hip.zmetric(I0c, I1c, f01, -1.9)) and torch.equal(r["z1"], hip.zmetric(I1c, I0c, f10, -1.8))
    fb0 = hip.bwarp_tscaled(f10, f01, t4, "t", "1-t", withmask=bool(it & 1)); fb1 = hip.bwarp_tscaled(f01, f10, t4, "1-t", "t", withmask=bool(it & 1))
    ok = ok and torch.equal(r["flowback_0"], fb0) and torch.equal(r["flowback_1"], fb1)
    ok = ok and torch.equal(r["im0_tot"], hip.bwarp(I0c, fb0, bool(it & 1))) and torch.equal(r["im1_tot"], hip.bwarp(I1c, fb1, bool(it & 1)))
    # the splat run on the bounds table made from the low-resolution flow equals the exact-bounds splat up to summation order
    for img, flow, z, lo2, sm in ((I0, r["flow_t0"], r["z0"], lo[:, 2:], 1), (I1, r["flow_t1"], r["z1"], lo[:, :2], 2)):
        ws = hip.splat_bounds_upsampled(lo2, t4, sm, float(up), H, W)
        e1 = (hip.softsplat_fused(img, flow, z, "softmax", bounds_ws=ws) - hip.softsplat_fused(img, flow, z, "softmax", kernel="tile")).abs().max().item()
        ok = ok and e1 <= 4e-6
    bad += not ok
    if not ok: print("prep MISMATCH", N, h, w, up)
print("level0_prep vs unfused kernels: 20 shapes,", bad, "mismatches", flush=True)

# ---- split-packed 3x3 conv vs the register-staged split conv (bit identity), random shapes / sources / options ----
bad = 0
for it in range(40):
    N = random.choice([1, 1, 2]); H = random.choice([2, 4, 8, 10, 18, 34, 66]); W = random.choice([2, 4, 16, 32, 34, 62, 66, 130])
    nsrc = random.choice([1, 1, 2, 3]); cs = [random.choice([8, 16, 24, 48]) for _ in range(nsrc - 1)] + [random.choice([1, 3, 4, 8, 13, 48, 52])]
    if sum(cs) > 112: cs = [48, 48, 4]
    ups = [random.random() < 0.3 for _ in cs]
    cout = random.choice([1, 4, 6, 16, 17, 32, 40, 48, 64, 96]); cst = random.choice([None, None, min(cout, 4)])
    relu = bool(it & 1); res = random.random() < 0.3
    srcs = [torch.randn(N, c, H // 2 if u else H, W // 2 if u else W, device=dev) for c, u in zip(cs, ups)]
    wt = torch.randn(cout, sum(cs), 3, 3, device=dev) / 20; b = torch.randn(cout, device=dev)
    rs = torch.randn(N, cst or cout, H, W, device=dev) if res else None
    ref = hip.conv2d(srcs, wt, b, relu=relu, residual=rs, cout_store=cst, up2=ups, precision="split")
    for su in (-1, 96):
        hip.lib().fldr_debug_spk_small_units(su)
        for variant, cons, tw in ((0, 8, 0), (1, 4, 0), (1, 8, 32), (1, 8, 16), (1, 8, 0)):   # barrier pipeline; ring with 4 consumers, 8 on 8x32 / 8x16 tiles / picked per launch
            hip.lib().fldr_debug_spk_variant(variant); hip.lib().fldr_debug_ring_consumers(cons); hip.lib().fldr_debug_ring_tile_width(tw)
            got, gp = hip.conv2d_spk(srcs, wt, b, relu=relu, residual=rs, cout_store=cst, up2=ups, want_f32=True, want_spk=True)
            ok = torch.equal(ref, got) and torch.equal(hip.spk_pack(ref).buf, gp.buf)
            bad += not ok
            if not ok: print("spk MISMATCH", N, cs, ups, cout, cst, H, W, relu, res, "small_units", su, "variant", variant, cons, tw)
hip.lib().fldr_debug_spk_small_units(96); hip.lib().fldr_debug_spk_variant(1); hip.lib().fldr_debug_ring_consumers(8); hip.lib().fldr_debug_ring_tile_width(0)
bad += hip.lib().fldr_debug_ring_timeouts() != 0
print("split-packed conv (3 pipelines) vs split conv: 40 shapes x 2 unit policies,", bad, "mismatches", flush=True)
bad = 0
for it in range(30):          # 32x32x16 ring kernel (64 / 96 outputs, packed output) vs the 16x16x32 ring: fp32 rounding apart, no ring timeouts
    N = random.choice([1, 1, 2, 3]); cout = random.choice([64, 96])
    cs = random.choice([[96], [48, 48, 4], [32, 32], [64, 32], [16], [40], [8, 8, 3]]); ups = [random.random() < 0.3 for _ in cs]
    H = 2 * random.choice([2, 3, 5, 9, 17, 33, 65]); W = 2 * random.choice([2, 3, 6, 32, 33, 34, 65, 67, 129])
    srcs = [torch.randn(N, c, H // (2 if u else 1), W // (2 if u else 1), device=dev) for c, u in zip(cs, ups)]
    if any(c % 8 for c in cs[:-1]): continue
    packed = [hip.spk_pack(x) for x in srcs]
    wt = torch.randn(cout, sum(cs), 3, 3, device=dev) / (sum(cs) * 9) ** 0.5; b = torch.randn(cout, device=dev)
    outs = []
    hip.lib().fldr_debug_spk_small_units(-1)
    for r32 in (0, 2):
        hip.lib().fldr_debug_ring32(r32)
        outs.append(hip.conv2d_spk(packed, wt, b, relu=bool(it & 1), up2=ups, want_f32=False, want_spk=True).float())
    hip.lib().fldr_debug_ring32(1); hip.lib().fldr_debug_spk_small_units(96)
    tol = 3e-6 * float(outs[0].abs().max()) + 1e-7
    ok = (outs[0] - outs[1]).abs().max().item() <= tol
    bad += not ok
    if not ok: print("ring32 MISMATCH", N, cs, ups, cout, H, W, (outs[0] - outs[1]).abs().max().item(), tol)
bad += hip.lib().fldr_debug_ring_timeouts() != 0
print("32x32x16 ring conv vs 16x16x32 ring: 30 shapes,", bad, "mismatches", flush=True)

# ---- one-pass PCA (incl. the 4-components-per-thread variant on small grids) vs the two-pass kernel ----
bad = 0
for it in range(16):
    P = random.choice([6, 12]); K = random.choice([4, 8, 16]); H = 8 * random.choice([1, 2, 9, 18, 36, 40]); W = 8 * random.choice([1, 3, 15, 30, 64, 70])
    ev = torch.randn(K, 64, device=dev, dtype=torch.float64); mean = torch.randn(64, device=dev, dtype=torch.float64) * 0.1
    mv = torch.rand(K, device=dev, dtype=torch.float64) + 0.5
    pl = torch.rand(P, H, W, device=dev) * 2 - 1
    o32, o64, mm = hip.pca_project(pl, ev, mean, mv, want_f64=True)
    s32, s64, smm, spk = hip.pca_project_stream(pl, ev, mean, mv, want_spk=True)
    ok = torch.equal(o32, s32) and torch.equal(o64, s64) and torch.equal(mm, smm) and torch.equal(hip.spk_pack(o32.view(1, P * K, H // 8, W // 8)).buf, spk.buf)
    bad += not ok
    if not ok: print("pca MISMATCH", P, K, H, W)
print("one-pass PCA vs two-pass: 16 shapes,", bad, "mismatches", flush=True)

# ---- fused dec3 + tail vs conv + synth_tail ----
bad = 0
for it in range(12):
    N = random.choice([1, 2]); h = random.choice([1, 2, 7, 8, 9, 20]); w = random.choice([1, 2, 31, 32, 33, 70])
    d2 = torch.rand(N, 16, h, w, device=dev); wt = torch.randn(6, 16, 3, 3, device=dev) / 6; bs = torch.randn(6, device=dev) * 0.3
    x = torch.rand(N, 3, 2, 2 * h, 2 * w, device=dev) * 2 - 1
    cands = [torch.rand(N, 3, 2 * h, 2 * w, device=dev) * 2 - 1 for _ in range(4)] + [x[:, :, 0], x[:, :, 1]]
    t = torch.rand(N, 1, device=dev)
    out = hip.dec3_synth(d2, wt, bs, cands, t, 1.5616)
    unf = hip.synth_tail(hip.conv2d([d2], wt, bs, up2=[True]), cands, t, 1.5616)
    err = (out - unf).abs().max().item()
    if not err < 3e-6: bad += 1; print("dec3 MISMATCH", N, h, w, err)
print("fused dec3+tail vs conv + synth_tail: 12 shapes,", bad, "mismatches", flush=True)

# ---- round 3: fused dec3 + blend on the matrix cores (packed source) vs the fp32-FMA kernel ----
bad = 0
for it in range(14):
    N = random.choice([1, 2, 3]); h = random.choice([1, 2, 7, 8, 9, 20, 41]); w = random.choice([1, 2, 15, 31, 32, 33, 70])
    d2 = torch.rand(N, 16, h, w, device=dev) * random.choice([0.3, 1.0, 4.0]); wt = torch.randn(6, 16, 3, 3, device=dev) / 6; bs = torch.randn(6, device=dev) * 0.3
    x = torch.rand(N, 3, 2, 2 * h, 2 * w, device=dev) * 2 - 1
    cands = [torch.rand(N, 3, 2 * h, 2 * w, device=dev) * 2 - 1 for _ in range(4)] + [x[:, :, 0], x[:, :, 1]]
    t = torch.rand(N, 1, device=dev)
    vec = hip.dec3_synth(d2, wt, bs, cands, t, 1.5616)
    mat = hip.dec3_synth(hip.spk_pack(d2), wt, bs, cands, t, 1.5616)
    err = (mat - vec).abs().max().item()
    if not err < 5e-6: bad += 1; print("dec3 matrix-core MISMATCH", N, h, w, err)
print("dec3+blend on the matrix cores vs the fp32-FMA kernel: 14 shapes,", bad, "mismatches", flush=True)

# ---- round 3: image splat, walk by runs of four pixels vs one pixel per item (same fp32 products, fp64 sums) ----
bad = 0
for it in range(24):
    N = random.choice([1, 2]); C = random.choice([1, 2, 3]); H = random.choice([1, 3, 11, 24, 25, 49, 70]); W = 4 * random.choice([1, 2, 13, 16, 17, 33, 75])
    mode = random.choice(["summation", "average", "linear", "softmax"])
    x = torch.rand(N, C, H, W, device=dev) * 2 - 1
    flow = (torch.rand(N, 2, H, W, device=dev) - 0.5) * random.choice([0.0, 1.0, 8.0, 100.0])
    if random.random() < 0.6:
        flow = torch.nn.functional.interpolate(torch.randn(N, 2, 2, 3, device=dev) * random.choice([1.0, 5.0, 40.0]), size=(H, W), mode="bilinear").contiguous()
    z = [torch.rand(N, 1, H, W, device=dev) + 0.1] if mode in ("linear", "softmax") else None
    outs = []
    for q in (1, 0):
        hip.lib().fldr_debug_splat_quad(q)
        outs.append(hip.softsplat_acc64([x], [flow], z, mode)[0])
    d = (outs[0] - outs[1]).abs()
    tol = 1e-5 * max(1.0, float(outs[1].abs().max())) if mode == "summation" else 2.5e-7
    if not (float(d.max()) <= tol): bad += 1; print("acc64 run-walk MISMATCH", N, C, H, W, mode, float(d.max()))
hip.lib().fldr_debug_splat_quad(1)
print("image splat by runs of four vs one pixel per item: 24 shapes,", bad, "mismatches", flush=True)

# ---- round 3: PCA pyramid with parked projections (every level / none) ----
bad = 0
for it in range(8):
    K = random.choice([4, 8, 16]); P = random.choice([5, 6, 12])
    ev = torch.randn(K, 64, device=dev, dtype=torch.float64); mean = torch.randn(64, device=dev, dtype=torch.float64) * 0.1
    mv = torch.rand(K, device=dev, dtype=torch.float64) + 0.5
    levels = [(8 * random.choice([1, 2, 9, 18, 36]), 8 * random.choice([1, 3, 15, 30, 64])) for _ in range(random.choice([1, 2, 4, 6]))]
    planes = [torch.rand(P, h, w, device=dev) * 2 - 1 for (h, w) in levels]
    a32, asp, amm = hip.pca_project_pyramid(planes, ev, mean, mv, want_f32=True, want_spk=True, raw_min_bytes=1 << 40)
    b32, bsp, bmm = hip.pca_project_pyramid(planes, ev, mean, mv, want_f32=True, want_spk=True, raw_min_bytes=0)
    ok = torch.equal(amm, bmm) and all(torch.equal(x, y) for x, y in zip(a32, b32)) and all(torch.equal(x.buf, y.buf) for x, y in zip(asp, bsp))
    bad += not ok
    if not ok: print("pca parked MISMATCH", K, P, levels)
print("PCA pyramid, projections parked vs recomputed: 8 pyramids,", bad, "mismatches", flush=True)
