"""Every stage of the 4K forward as the VICTIM on one HIP stream beside every stage as the PARTNER on a second stream (different frame pairs):
is the victim's output the same bits as when it runs alone?  The whole-forward check (tools/concurrency_check.py) samples these overlaps at
random; this one walks the matrix.   python tools/pairwise_concurrency.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn, fldr_hip as hip, pca_comp
dev = torch.device("cuda:0")
m, _, a = Hn.prepare_model(dev)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
t = torch.tensor([[0.5]], device=dev)
t4 = t.view(1, 1, 1, 1).float()
T, za0, za1 = m.vfinet._host_scalars()
n_levels, i8 = a.S_tst + 1, a.scales.index(8)
unet = m.vfinet.refine_unet
def flat(x):
    out = []
    def rec(y):
        if torch.is_tensor(y): out.append(y)
        elif isinstance(y, hip.Spk): out.append(y.buf)
        elif isinstance(y, (list, tuple)): [rec(z) for z in y]
        elif isinstance(y, dict): [rec(v) for k, v in sorted(y.items()) if not k.startswith("_")]
    rec(x); return out
def stages_for(seed):
    """-> dict name -> zero-argument function of the stage on this pair's tensors (inputs prepared here, one at a time)."""
    fr = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=seed)).to(dev)
    with torch.no_grad():
        pyr = Hn.build_pyramid(Hn.pad_frames(fr, a), a)
        H, W = pyr[0].shape[3:]
        f_pca = lambda: pca_comp.to_pca_diff_f32_pyramid([pyr[i].reshape(6, pyr[i].shape[3], pyr[i].shape[4]) for i in range(n_levels)], m.params, a, m.pca_means[i8], m.EVs[i8], m.mean_vecs[i8], want_spk=True, want_f32=True)
        pv, pp = f_pca()
        P = [hip.Spk(pp[i].buf, (1, 96, pv[i].shape[-2], pv[i].shape[-1])) for i in range(n_levels)]
        V = [pv[i].view(1, 96, pv[i].shape[-2], pv[i].shape[-1]) for i in range(n_levels)]
        c0, c2 = m.rec_ctx_ds[0], m.rec_ctx_ds[2]
        def f_feats():
            ys = hip.conv2d_spk_levels(P, c0.weight, c0.bias, relu=True, want_f32=False, want_spk=True)
            return hip.conv2d_spk_levels(ys, c2.weight, c2.bias, relu=True, residuals=V, want_f32=True, want_spk=True)
        feats = f_feats()
        def f_coarse():
            flow = None
            for lv in range(a.S_tst, 0, -1): flow = m.vfinet.estimate_flow(feats[lv], flow)
            return flow
        flow1 = f_coarse()
        f_flow0 = lambda: m.vfinet.estimate_flow(feats[0], flow1)
        flow0 = f_flow0()
        I0, I1 = pyr[0][:, :, 0], pyr[0][:, :, 1]
        def f_prep():
            r = hip.level0_prep(flow0, I0, I1, t4, H, W, za0, za1, withmask=True, want_z=True)
            return {k: v for k, v in r.items() if not k.startswith("_")}
        pre = f_prep()
        def f_splat():
            bw = hip.splat_bounds_upsampled_pair(flow0, t4, "images", 8, H, W)
            return hip.softsplat_acc64([I0, I1], [pre["flow_t0"], pre["flow_t1"]], [pre["z0"], pre["z1"]], "softmax", bounds_ws=bw)
        wp = f_splat()
        srcs = [I0, I1, wp[0], wp[1], pre["flow_t0"], pre["flow_t1"], pre["flowback_0"], pre["flowback_1"], pre["im0_tot"], pre["im1_tot"]]
        f_enc1 = lambda: hip.conv2d(srcs, unet.enc1.weight, unet.enc1.bias, stride=2, relu=True, want_f32=False, want_spk=True)
        enc1p = f_enc1()
        f_enc2 = lambda: hip.conv2d_s2_spk(enc1p, unet.enc2.weight, unet.enc2.bias, relu=True, want_f32=False, want_spk=True)
        enc2p = f_enc2()
        f_enc3 = lambda: hip.conv2d_s2_spk_pair(enc2p, unet._enc3_halves(), relu=True)
        e3 = f_enc3()
        f_dec0 = lambda: hip.conv2d_spk(e3, unet.dec0.weight, unet.dec0.bias, relu=True, want_f32=False, want_spk=True)
        d0 = f_dec0()
        f_dec1 = lambda: hip.conv2d_spk([d0, enc2p], unet.dec1.weight, unet.dec1.bias, relu=True, up2=[True, False], want_f32=False, want_spk=True)
        d1 = f_dec1()
        cands = [wp[0], wp[1], pre["im0_tot"], pre["im1_tot"], I0, I1]
        f_dec23 = lambda: hip.dec23_synth(d1, enc1p, unet.dec2.weight, unet.dec2.bias, unet.dec3.weight, unet.dec3.bias, cands, t4, T)
        f_dec23()
        # the two-kernel synthesis (FLDR_DEC23=0): dec2 as a ring convolution, then dec3 + softmax + blend
        d2 = unet.forward_until_dec2(srcs, packed_out=hip.DEC3_MFMA and hip.use_spk())
        f_dec3 = lambda: hip.dec3_synth(d2, unet.dec3.weight, unet.dec3.bias, cands, t4, T)
        f_dec3()
        torch.cuda.synchronize()
    keep = (fr, pyr, pv, pp, feats, flow1, flow0, pre, wp, enc1p, enc2p, e3, d0, d1, d2)
    return {"pca": f_pca, "rec_ctx_ds": f_feats, "flow levels 5-1": f_coarse, "flow level 0": f_flow0, "level0_prep": f_prep, "image splats": f_splat,
            "enc1": f_enc1, "enc2": f_enc2, "enc3": f_enc3, "dec0": f_dec0, "dec1": f_dec1, "dec23_synth": f_dec23, "dec3_synth (two-kernel path)": f_dec3}, keep
with torch.no_grad():
    A, keepA = stages_for(7)
    B, keepB = stages_for(8)
    alone = {}
    for name, fn in A.items():
        alone[name] = [x.clone() for x in flat(fn())]; torch.cuda.synchronize()
    # partners that are not stages: kernels that only keep the SIMDs busy (test build: csrc/test_partner_kernels.hip) — two workgroups per CU of
    # vector FMAs made round 6's defect show in 8 of 8 runs where the forward's own kernels needed dozens
    hog_out = torch.empty(1024 * 256, device=dev)
    for label, kind, wgs, lds, iters in (("busy: vector FMAs", 2, 512, 1024, 40000), ("busy: matrix instr.", 1, 512, 1024, 20000), ("busy: scalar adds", 3, 512, 1024, 40000),
                                         ("busy: LDS reads", 4, 512, 1024, 20000), ("busy: FMAs, 125 KB LDS", 2, 256, 125 * 1024, 40000)):
        B[label] = (lambda kind=kind, wgs=wgs, lds=lds, iters=iters: hip.busy_partner(hog_out, wgs, lds, iters, kind))
    sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    total = 0
    names = list(A)
    partners = list(B) if not os.environ.get("PARTNERS_BUSY_ONLY") else [n for n in B if n.startswith("busy")]
    print("%-18s %s" % ("victim \\ partner", " ".join("%6s" % n.replace("busy: ", "b:")[:6] for n in partners)))
    for v in names:
        row = []
        for p_ in partners:
            nbad = 0
            if os.environ.get("VERBOSE_CELLS"): print("    cell: %s beside %s" % (v, p_), flush=True)
            for rep in range(REPS):
                sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
                outs = []
                for j in range(3):
                    with torch.cuda.stream(sB): k1 = B[p_]()
                    with torch.cuda.stream(sA): outs.append(A[v]())
                    with torch.cuda.stream(sB): k2 = B[p_]()
                torch.cuda.synchronize()
                nbad += sum(1 for o in outs if not all(torch.equal(x, y) for x, y in zip(flat(o), alone[v])))
                del outs, k1, k2
            total += nbad
            row.append(nbad)
        print("%-18s %s" % (v, " ".join("%6d" % x for x in row)), flush=True)
    print("TOTAL victim results differing from the stage alone: %d (of %d per cell)" % (total, 3 * REPS))
sys.exit(1 if total else 0)
