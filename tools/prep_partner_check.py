"""level0_prep on one HIP stream beside ONE other kernel of the forward on a second stream: does prep's output equal its output when it runs alone?
(How the concurrency defect of round 6 was narrowed down: profiles/r06_prep_concurrency.txt.)   FLDR_LIB=<variant> python tools/prep_partner_check.py [reps]"""
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
