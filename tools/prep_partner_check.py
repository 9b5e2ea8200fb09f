"""level0_prep on one HIP stream beside ONE other kernel of the forward on a second stream: does prep's output equal its output when it runs alone?
(How the concurrency defect of round 6 was narrowed down: profiles/r06_prep_concurrency.txt.)   FLDR_LIB=<variant> python tools/prep_partner_check.py [reps]
HOG=1: the partners are busy-partner kernels of the test build instead (sleeping / matrix instructions / vector FMAs / scalar adds / LDS reads); HOG=footprints: the
matrix-instruction partner with ten register footprints; DUMP=1 | rows | hwid with the probe builds of tools/asm_edits/: where the outputs differ."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn, fldr_hip as hip, pca_comp
dev = torch.device("cuda:0")
m, _, a = Hn.prepare_model(dev)
t = torch.tensor([[0.5]], device=dev)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = [Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=p)).to(dev) for p in range(2)]
n_levels, i8 = a.S_tst + 1, a.scales.index(8)
unet = m.vfinet.refine_unet
t4 = t.view(1, 1, 1, 1).float()
T, za0, za1 = m.vfinet._host_scalars()
with torch.no_grad():
    pyrs = [Hn.build_pyramid(Hn.pad_frames(f, a), a) for f in frames]
    H, W = pyrs[0][0].shape[3:]
    flows = []
    for p in pyrs:
        pv, pp = pca_comp.to_pca_diff_f32_pyramid([p[i].reshape(6, p[i].shape[3], p[i].shape[4]) for i in range(n_levels)], m.params, a, m.pca_means[i8], m.EVs[i8], m.mean_vecs[i8], want_spk=True, want_f32=True)
        flow = None
        for lv in range(a.S_tst, -1, -1):
            h, w = p[lv].shape[3] // 8, p[lv].shape[4] // 8
            flow = m.vfinet.estimate_flow(m._extract_features(pv[lv].view(1, 96, h, w), hip.Spk(pp[lv].buf, (1, 96, h, w))), flow)
        flows.append(flow)
    def prep_of(k):
        r = hip.level0_prep(flows[k], pyrs[k][0][:, :, 0], pyrs[k][0][:, :, 1], t4, H, W, za0, za1, withmask=True, want_z=True)
        return {kk: v for kk, v in r.items() if not kk.startswith("_")}
    pre = [prep_of(k) for k in range(2)]
    torch.cuda.synchronize()
    I0, I1 = pyrs[0][0][:, :, 0], pyrs[0][0][:, :, 1]
    bw = hip.splat_bounds_upsampled_pair(flows[0], t4, "images", 8, H, W)
    wp = hip.softsplat_acc64([I0, I1], [pre[0]["flow_t0"], pre[0]["flow_t1"]], [pre[0]["z0"], pre[0]["z1"]], "softmax", bounds_ws=bw)
    srcs = [I0, I1, wp[0], wp[1], pre[0]["flow_t0"], pre[0]["flow_t1"], pre[0]["flowback_0"], pre[0]["flowback_1"], pre[0]["im0_tot"], pre[0]["im1_tot"]]
    enc1p = hip.conv2d(srcs, unet.enc1.weight, unet.enc1.bias, stride=2, relu=True, want_f32=False, want_spk=True)
    enc2p = hip.conv2d_s2_spk(enc1p, unet.enc2.weight, unet.enc2.bias, relu=True, want_f32=False, want_spk=True)
    e3 = hip.conv2d_s2_spk_pair(enc2p, unet._enc3_halves(), relu=True)
    d0 = hip.conv2d_spk(e3, unet.dec0.weight, unet.dec0.bias, relu=True, want_f32=False, want_spk=True)
    torch.cuda.synchronize()
    partners = {
        "enc1 (persistent stride-2)": lambda: hip.conv2d(srcs, unet.enc1.weight, unet.enc1.bias, stride=2, relu=True, want_f32=False, want_spk=True),
        "dec1 (ring, 125 KB LDS)": lambda: hip.conv2d_spk([d0, enc2p], unet.dec1.weight, unet.dec1.bias, relu=True, up2=[True, False], want_f32=False, want_spk=True),
        "dec0 (ring, 125 KB LDS)": lambda: hip.conv2d_spk(e3, unet.dec0.weight, unet.dec0.bias, relu=True, want_f32=False, want_spk=True),
        "image splats": lambda: hip.softsplat_acc64([I0, I1], [pre[0]["flow_t0"], pre[0]["flow_t1"]], [pre[0]["z0"], pre[0]["z1"]], "softmax", bounds_ws=bw),
    }
    if os.environ.get("HOG"):
        hog_out = torch.empty(1024 * 256, device=dev)
        def hog(wgs, lds, iters, kind):
            return lambda: hip.busy_partner(hog_out, wgs, lds, iters, kind)
        if os.environ.get("HOG") == "footprints":
            partners = {"busy (matrix instructions, 2 / CU), register footprint %d" % fp: hog(512, 1024, 20000, 1 + 16 * fp) for fp in range(10)}
        elif os.environ.get("DUMP"): partners = {"hog: 1 KB LDS, matrix instructions, 2 / CU": hog(512, 1024, 20000, 1)}
        else: partners = {
            "hog: 125 KB LDS, sleeping": hog(256, 125 * 1024, 20000, 0),
            "hog: 125 KB LDS, MFMAs": hog(256, 125 * 1024, 20000, 1),
            "hog: 1 KB LDS, MFMAs, 2 / CU": hog(512, 1024, 20000, 1),
            "hog: 1 KB LDS, vector FMAs, 2 / CU": hog(512, 1024, 40000, 2),
            "hog: 1 KB LDS, scalar adds, 2 / CU": hog(512, 1024, 40000, 3),
            "hog: 1 KB LDS, LDS reads, 2 / CU": hog(512, 1024, 20000, 4),
            "hog: 1 KB LDS, MFMAs, 1 wave per SIMD (256 WGs of 256)": hog(256, 1024, 20000, 1),
        }
    sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    total = 0
    for name, fn in partners.items():
        nbad = 0
        for rep in range(REPS):
            sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
            outs = []
            for j in range(4):
                with torch.cuda.stream(sB): keep = fn()
                with torch.cuda.stream(sA): outs.append(prep_of(1))
                with torch.cuda.stream(sB): keep2 = fn()
            torch.cuda.synchronize()
            nbad += sum(1 for o in outs if not all(torch.equal(o[kk], pre[1][kk]) for kk in o))
            if os.environ.get("DUMP") == "hwid":                        # probe5 build: flowback_0 = (HW_ID, XCC_ID) of the pixel's wave
                import collections
                o = outs[0]
                hw = o["flowback_0"][0, 0].view(torch.int32); xc = o["flowback_0"][0, 1].view(torch.int32)
                bad = (o["im1_tot"][0, 0] != pre[1]["im1_tot"][0, 0])
                badw = bad.view(bad.shape[0], -1, 64)[:, :, 63]              # per wave (lane 63)
                hww = hw.view(hw.shape[0], -1, 64)[:, :, 63]; xcw = xc.view(xc.shape[0], -1, 64)[:, :, 63]
                def field(x, lo, n): return (x >> lo) & ((1 << n) - 1)
                for nm, val in (("wave_id", field(hww, 0, 4)), ("simd_id", field(hww, 4, 2)), ("pipe_id", field(hww, 6, 2)), ("cu_id", field(hww, 8, 4)), ("sh_id", field(hww, 12, 1)),
                                ("se_id", field(hww, 13, 3)), ("tg_id", field(hww, 16, 4)), ("queue_id", field(hww, 24, 3)), ("xcc_id", field(xcw, 0, 4))):
                    allc = collections.Counter(val.flatten().tolist()); badc = collections.Counter(val[badw].flatten().tolist())
                    print("      %-8s failing / all waves: %s" % (nm, ["%d: %d/%d" % (k, badc.get(k, 0), allc[k]) for k in sorted(allc)]), flush=True)
                print("      failing waves %d of %d" % (int(badw.sum()), badw.numel()), flush=True)
            elif os.environ.get("DUMP"):                                  # experiment builds (PREP_DUMP): which plane differs where, first few values
                for o in outs:
                    for kk in o:
                        ne = (o[kk] != pre[1][kk])
                        if ne.any():
                            for ch in range(o[kk].shape[1]):
                                nz = ne[0, ch].nonzero()
                                if len(nz):
                                    if os.environ.get("DUMP") == "rows":
                                        dl = (pre[1][kk][0, ch] - o[kk][0, ch])
                                        rows = sorted(set(nz[:, 0].tolist()))
                                        print("      %s[%d] alone - beside by row: %s" % (kk, ch, [(r, sorted(set(dl[r][ne[0, ch, r]].tolist()))[:3]) for r in rows[:6] + rows[len(rows) // 2:len(rows) // 2 + 3]]), flush=True)
                                        # which waves: (row % 4 = wave of the workgroup, column block)
                                        import collections
                                        print("      rows mod 4: %s; differing 16-pixel runs per row (first rows): %s" % (sorted(collections.Counter((nz[:, 0] % 4).tolist()).items()), [(r, int(ne[0, ch, r].sum()) // 16) for r in rows[:8]]), flush=True)
                                    y, x = int(nz[0, 0]), int(nz[0, 1])
                                    print("      %s[%d]: %d differ, lanes %s; first (%d,%d): alone %r beside %r" % (kk, ch, len(nz), sorted(set((nz[:, 1] % 64).tolist()))[:4] + ["..."] + sorted(set((nz[:, 1] % 64).tolist()))[-2:],
                                          y, x, pre[1][kk][0, ch, y, x:x + 3].tolist(), o[kk][0, ch, y, x:x + 3].tolist()), flush=True)
                    break
            if os.environ.get("VERBOSE"):
                for o in outs:
                    d = {kk: int((o[kk] != pre[1][kk]).sum()) for kk in o if not torch.equal(o[kk], pre[1][kk])}
                    if d:
                        kk = sorted(d)[0]
                        nz = (o[kk] != pre[1][kk]).nonzero()
                        print("      differing outputs %s; %s: lanes %s, rows %d-%d" % (d, kk, sorted(set((nz[:, 3] % 64).tolist()))[:20], int(nz[:, 2].min()), int(nz[:, 2].max())), flush=True)
                        break
        total += nbad
        print("prep beside %-28s: %d of %d prep results differ from prep alone" % (name, nbad, 4 * REPS), flush=True)
sys.exit(1 if total else 0)
