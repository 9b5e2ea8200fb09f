"""Where does an 8x multi-t pair (4096x2160, 7 outputs, pair-invariant cache) spend its time — host enqueue or GPU?
Per repetition: wall time of the call (host enqueue), wall time to the synchronised end, and per output the GPU time between events."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
pairs = [Hn.frames_from_uint8(Hn.synthetic_pair(2160, 4096, seed=200 + k)).to(dev) for k in range(2)]
ts = [k / 8 for k in range(1, 8)]
with torch.no_grad():
    pyrs = [Hn.build_pyramid(Hn.pad_frames(f, args), args) for f in pairs]
    for mode in ("cache", "nocache", "cache"):
        for rep in range(6):
            k = rep % 2
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(ts) + 1)]
            t0 = time.perf_counter()
            if mode == "cache":
                model.pair_cache = True
            host = []
            ev[0].record()
            outs = []
            for i, tv in enumerate(ts):
                h0 = time.perf_counter()
                t = torch.full((1, 1), float(tv), device=dev)
                outs.append(model([None] * 6, t, normInput=pyrs[k], is_training=False, validation=False)[0])
                ev[i + 1].record()
                host.append((time.perf_counter() - h0) * 1e3)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            model.pair_cache = False
            model._pair_state = None
            gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(ts))]
            print("%-7s rep %d: enqueue %.2f ms, total %.2f ms | host per output %s | gpu per output %s" % (
                mode, rep, (t1 - t0) * 1e3, (t2 - t0) * 1e3, " ".join("%.2f" % x for x in host), " ".join("%.2f" % x for x in gpu)), flush=True)
            del outs
