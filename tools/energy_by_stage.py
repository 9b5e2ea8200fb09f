#!/usr/bin/env python3
"""Energy of one 3840x2160 forward by STAGE (GPU box; driven by tools/energy_table.sh, which samples rocm-smi meanwhile).

Every stage of the forward — the same Python-level calls DCTXVFInet.forward makes, on the tensors of a real forward of a synthetic 4K
pair — is run back to back for a few seconds; this process prints `STAGE <name> <t_start> <t_end> <us per pass> <launches>` with wall-clock
stamps, the shell samples board power and shader clock with time stamps, and `--join` folds the two into
    stage | us per forward | W while it loops | GHz | mJ per forward (W x us) | dynamic mJ ((W - idle W) x us) | share
plus the sustained three-pairs-in-flight bench loop (W, GHz, ms per step -> J per pair) as the last stage: the binding roofline of this
path is the board's power cap, and this table is its evidence.

    python tools/energy_by_stage.py run [seconds per stage]        # prints STAGE lines (stdout)
    python tools/energy_by_stage.py join <stages.log> <samples.log> <out.txt> [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def run(secs):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "fldr-vfi_amd"))
    import fldr_hip as hip
    import fldr_harness as Hn
    import pca_comp
    dev = torch.device("cuda:0")
    m, _, a = Hn.prepare_model(dev)
    vfi, un = m.vfinet, m.vfinet.refine_unet
    Hs, Ws = 2160, 3840
    pairs = [Hn.frames_from_uint8(Hn.synthetic_pair(Hs, Ws, seed=s, quadrant=bool(s & 1))).to(dev) for s in range(3)]
    pyrs = [Hn.build_pyramid(Hn.pad_frames(f, a), a) for f in pairs]
    H, W = pyrs[0][0].shape[3:]
    t = torch.tensor([[0.5]], device=dev)
    t4 = t.view(1, 1, 1, 1).float()
    T, za0, za1 = vfi._host_scalars()
    stages = []

    def stage(name, fn, launches):
        stages.append((name, fn, launches))

    with torch.no_grad():
        # ---- tensors of a real forward, per pair (rotated so that no stage re-reads a cache-resident input)
        st = []
        for pyr in pyrs:
            d = {"pyr": pyr}
            planes = [pyr[i].reshape(6, pyr[i].shape[3], pyr[i].shape[4]) for i in range(6)]
            d["planes"] = planes
            pcas, pcs = pca_comp.to_pca_diff_f32_pyramid(planes, m.params, a, m.Mean8, m.EV8, m.meanVec8, want_spk=True, want_f32=True)
            d["pv"] = [pcas[i].view(1, 96, pyr[i].shape[3] // 8, pyr[i].shape[4] // 8) for i in range(6)]
            d["pp"] = [hip.Spk(pcs[i].buf, (1, 96, pyr[i].shape[3] // 8, pyr[i].shape[4] // 8)) for i in range(6)]
            c0, c2 = m.rec_ctx_ds[0], m.rec_ctx_ds[2]
            ys = hip.conv2d_spk_levels(d["pp"], c0.weight, c0.bias, relu=True, want_f32=False, want_spk=True)
            d["feats"] = hip.conv2d_spk_levels(ys, c2.weight, c2.bias, relu=True, residuals=d["pv"], want_f32=True, want_spk=True)
            flows = {}
            flow = None
            for level in range(5, -1, -1):
                flow = vfi.estimate_flow(d["feats"][level], flow)
                flows[level] = flow
            d["flows"] = flows
            I0, I1 = pyr[0][:, :, 0], pyr[0][:, :, 1]
            r = hip.level0_prep(flows[0], I0, I1, t4, H, W, za0, za1, withmask=True, want_z=True)
            bw = hip.splat_bounds_upsampled_pair(flows[0], t4, "images", 8, H, W)
            w0, w1 = hip.softsplat_acc64([I0, I1], [r["flow_t0"], r["flow_t1"]], [r["z0"], r["z1"]], "softmax", bounds_ws=bw)
            d.update(I0=I0, I1=I1, r=r, w0=w0, w1=w1)
            d["srcs"] = [I0, I1, w0, w1, r["flow_t0"], r["flow_t1"], r["flowback_0"], r["flowback_1"], r["im0_tot"], r["im1_tot"]]
            d["cands"] = [w0, w1, r["im0_tot"], r["im1_tot"], I0, I1]
            d["enc1p"] = hip.conv2d(d["srcs"], un.enc1.weight, un.enc1.bias, stride=2, relu=True, want_f32=False, want_spk=True)
            d["enc2p"] = hip.conv2d_s2_spk(d["enc1p"], un.enc2.weight, un.enc2.bias, relu=True, want_f32=False, want_spk=True)
            d["enc3p"] = hip.conv2d_s2_spk_pair(d["enc2p"], un._enc3_halves(), relu=True)
            d["dec0p"] = hip.conv2d_spk(d["enc3p"], un.dec0.weight, un.dec0.bias, relu=True, want_f32=False, want_spk=True)
            d["dec1p"] = hip.conv2d_spk([d["dec0p"], d["enc2p"]], un.dec1.weight, un.dec1.bias, relu=True, up2=[True, False], want_f32=False, want_spk=True)
            st.append(d)
        torch.cuda.synchronize()
        c0, c2 = m.rec_ctx_ds[0], m.rec_ctx_ds[2]

        def s_pca(i):
            d = st[i % 3]
            pca_comp.to_pca_diff_f32_pyramid(d["planes"], m.params, a, m.Mean8, m.EV8, m.meanVec8, want_spk=True, want_f32=True)

        def s_rec(i):
            d = st[i % 3]
            ys = hip.conv2d_spk_levels(d["pp"], c0.weight, c0.bias, relu=True, want_f32=False, want_spk=True)
            hip.conv2d_spk_levels(ys, c2.weight, c2.bias, relu=True, residuals=d["pv"], want_f32=True, want_spk=True)

        def s_coarse(i):
            d = st[i % 3]
            flow = None
            for level in range(5, 1, -1):
                flow = vfi.estimate_flow(d["feats"][level], flow)

        def s_l1(i):
            d = st[i % 3]
            vfi.estimate_flow(d["feats"][1], d["flows"][2])

        def s_l0(i):
            d = st[i % 3]
            vfi.estimate_flow(d["feats"][0], d["flows"][1])

        def s_prep(i):
            d = st[i % 3]
            hip.level0_prep(d["flows"][0], d["I0"], d["I1"], t4, H, W, za0, za1, withmask=True, want_z=True)

        def s_splat(i):
            d = st[i % 3]
            bw = hip.splat_bounds_upsampled_pair(d["flows"][0], t4, "images", 8, H, W)
            hip.softsplat_acc64([d["I0"], d["I1"]], [d["r"]["flow_t0"], d["r"]["flow_t1"]], [d["r"]["z0"], d["r"]["z1"]], "softmax", bounds_ws=bw)

        def s_enc1(i):
            d = st[i % 3]
            hip.conv2d(d["srcs"], un.enc1.weight, un.enc1.bias, stride=2, relu=True, want_f32=False, want_spk=True)

        def s_enc23(i):
            d = st[i % 3]
            e2 = hip.conv2d_s2_spk(d["enc1p"], un.enc2.weight, un.enc2.bias, relu=True, want_f32=False, want_spk=True)
            hip.conv2d_s2_spk_pair(e2, un._enc3_halves(), relu=True)

        def s_dec01(i):
            d = st[i % 3]
            o = hip.conv2d_spk(d["enc3p"], un.dec0.weight, un.dec0.bias, relu=True, want_f32=False, want_spk=True)
            hip.conv2d_spk([o, d["enc2p"]], un.dec1.weight, un.dec1.bias, relu=True, up2=[True, False], want_f32=False, want_spk=True)

        def s_dec23(i):
            d = st[i % 3]
            hip.dec23_synth(d["dec1p"], d["enc1p"], un.dec2.weight, un.dec2.bias, un.dec3.weight, un.dec3.bias, d["cands"], t4, T)

        def s_forward(i):
            Hn.interpolate(m, a, pairs[i % 3], t, pyramid=pyrs[i % 3])

        stage("pca (3 launches)", s_pca, 3)
        stage("rec_ctx_ds, all levels (2)", s_rec, 2)
        stage("flow levels 5-2 (32)", s_coarse, 32)
        stage("flow level 1 (8)", s_l1, 8)
        stage("flow level 0 (8)", s_l0, 8)
        stage("level0_prep (2)", s_prep, 2)
        stage("image splats + bounds (2)", s_splat, 2)
        stage("enc1 (1)", s_enc1, 1)
        stage("enc2 + enc3 (2)", s_enc23, 2)
        stage("dec0 + dec1 (2)", s_dec01, 2)
        stage("dec23_synth (1)", s_dec23, 1)
        stage("whole forward, one stream (60)", s_forward, 60)

        print("STAGE idle %.3f %.3f 0 0" % (time.time(), time.time() + secs), flush=True)
        time.sleep(secs)
        for name, fn, launches in stages:
            for i in range(6):
                fn(i)
            torch.cuda.synchronize()
            n = 0
            t0 = time.perf_counter()
            w0 = time.time()
            while time.perf_counter() - t0 < secs:
                for i in range(10):
                    fn(n + i)
                n += 10
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("STAGE %s | %.3f %.3f %.2f %d" % (name, w0, time.time(), dt / n * 1e6, launches), flush=True)
        # the sustained loop of the bench: three pairs in flight on three streams
        streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        cur = torch.cuda.current_stream(dev)

        def sustained(seconds):
            n = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                for k in range(12):
                    s = streams[k % 3]
                    with torch.cuda.stream(s):
                        Hn.interpolate(m, a, pairs[k % 3], t, pyramid=pyrs[k % 3])
                n += 12
                for s in streams:
                    s.synchronize()
            return n, time.perf_counter() - t0
        sustained(1.0)
        w0 = time.time()
        n, dt = sustained(max(secs, 6.0))
        print("STAGE sustained loop, 3 pairs in flight | %.3f %.3f %.2f 60" % (w0, time.time(), dt / n * 1e6), flush=True)
        hip.check_range()
    print("DONE", flush=True)


def join(stages_log, samples_log, out_txt, out_json=None):
    samples = []
    for line in open(samples_log):
        p = line.split()
        if len(p) >= 3:
            try:
                samples.append((float(p[0]), float(p[1]), float(p[2])))
            except ValueError:
                pass
    rows = []
    for line in open(stages_log):
        if not line.startswith("STAGE "):
            continue
        body = line[6:].strip()
        if "|" in body:
            name, rest = body.split("|")
            p = rest.split()
        else:
            p = body.split()
            name, p = p[0], p[1:]
        t0, t1, us, launches = float(p[0]), float(p[1]), float(p[2]), int(p[3])
        sel = [s for s in samples if t0 + 1.0 <= s[0] <= t1 - 0.2]          # (settling second dropped)
        W = sum(s[1] for s in sel) / max(len(sel), 1)
        mhz = sum(s[2] for s in sel) / max(len(sel), 1)
        rows.append(dict(stage=name.strip(), us=us, watts=W, mhz=mhz, samples=len(sel), launches=launches))
    idle = next((r["watts"] for r in rows if r["stage"] == "idle"), 0.0)
    parts = [r for r in rows if r["stage"] not in ("idle",) and not r["stage"].startswith("whole forward") and not r["stage"].startswith("sustained")]
    whole = next((r for r in rows if r["stage"].startswith("whole forward")), None)
    sust = next((r for r in rows if r["stage"].startswith("sustained")), None)
    for r in rows:
        r["mJ"] = r["watts"] * r["us"] * 1e-3
        r["mJ_dynamic"] = (r["watts"] - idle) * r["us"] * 1e-3
    tot_us = sum(r["us"] for r in parts)
    tot_dyn = sum(r["mJ_dynamic"] for r in parts)
    lines = []
    lines.append("Energy of one 3840x2160 forward by stage (tools/energy_table.sh; each stage looped back to back on the tensors of a real forward,")
    lines.append("board power and shader clock from rocm-smi while it loops; idle board %.0f W).  mJ = W x us; dynamic mJ = (W - idle) x us." % idle)
    lines.append("")
    lines.append("%-40s %10s %8s %7s %9s %11s %7s" % ("stage (launches)", "us / fwd", "W", "GHz", "mJ", "dynamic mJ", "share"))
    for r in sorted(parts, key=lambda r: -r["mJ_dynamic"]):
        lines.append("%-40s %10.1f %8.0f %7.2f %9.1f %11.1f %6.1f%%" % (r["stage"], r["us"], r["watts"], r["mhz"] / 1e3, r["mJ"], r["mJ_dynamic"],
                                                                       100.0 * r["mJ_dynamic"] / max(tot_dyn, 1e-9)))
    lines.append("%-40s %10.1f %8s %7s %9.1f %11.1f" % ("sum of the stages", tot_us, "", "", sum(r["mJ"] for r in parts), tot_dyn))
    if whole:
        lines.append("%-40s %10.1f %8.0f %7.2f %9.1f %11.1f" % (whole["stage"], whole["us"], whole["watts"], whole["mhz"] / 1e3, whole["mJ"], whole["mJ_dynamic"]))
    summary = {}
    if sust:
        model = tot_dyn + idle * sust["us"] * 1e-3
        lines.append("%-40s %10.1f %8.0f %7.2f %9.1f %11.1f" % (sust["stage"] + " (per pair)", sust["us"], sust["watts"], sust["mhz"] / 1e3, sust["mJ"], sust["mJ_dynamic"]))
        lines.append("")
        lines.append("Sustained: %.3f ms per pair at %.0f W = %.3f J per pair.  Model: sum of the stages' dynamic energy %.3f J + idle %.0f W x step = %.3f J (%.1f %% of measured)."
                     % (sust["us"] * 1e-3, sust["watts"], sust["mJ"] * 1e-3, tot_dyn * 1e-3, idle, model * 1e-3, 100.0 * model / max(sust["mJ"], 1e-9)))
        summary = dict(joules_per_pair=round(sust["mJ"] * 1e-3, 4), watts_sustained=round(sust["watts"], 1), ms_per_step=round(sust["us"] * 1e-3, 4),
                       mhz_sustained=round(sust["mhz"], 0), idle_watts=round(idle, 1), stages_dynamic_joules=round(tot_dyn * 1e-3, 4),
                       model_over_measured=round(model / max(sust["mJ"], 1e-9), 4))
    open(out_txt, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if out_json:
        json.dump(dict(rows=rows, summary=summary), open(out_json, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "join":
        join(*sys.argv[2:6])
    else:
        run(float(sys.argv[2]) if len(sys.argv) > 2 else 4.0)
