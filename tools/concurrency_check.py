"""Forwards of DIFFERENT frame pairs kept in flight on several HIP streams (bench.py's loop, eager and as hipGraph replays) against the same
forwards run one at a time: every frame must be the same bits.  Found in round 6: one packed-fp32 instruction of level0_prep that gfx950
executes wrongly in lanes 48-63 beside another stream's matrix instructions (profiles/r06_prep_concurrency.txt).   FH= FW= frame size, FLDR_CONV_PRECISION=fp16 the fp16 mode.   python tools/concurrency_check.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
t = torch.tensor([[0.5]], device=dev)
NS, NP = 3, 4
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
H, W = int(os.environ.get("FH", 2160)), int(os.environ.get("FW", 3840))
frames = [Hn.frames_from_uint8(Hn.synthetic_pair(H, W, seed=p)).to(dev) for p in range(NP)]
with torch.no_grad():
    pyrs = [Hn.build_pyramid(Hn.pad_frames(f, args), args) for f in frames]
    refs = [Hn.interpolate(model, args, frames[k], t, pyramid=pyrs[k]).clone() for k in range(NP)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
bad_total = 0
with torch.no_grad():
    for rep in range(REPS):
        for s in streams: s.wait_stream(torch.cuda.current_stream())
        outs = []
        for i in range(12):
            with torch.cuda.stream(streams[i % NS]):
                outs.append((i % NP, Hn.interpolate(model, args, frames[i % NP], t, pyramid=pyrs[i % NP])))
        torch.cuda.synchronize()
        bad = [(i, k, float((o - refs[k]).abs().max())) for i, (k, o) in enumerate(outs) if not torch.equal(o, refs[k])]
        bad_total += len(bad)
        print("eager, 3 streams, rep %d: %d of 12 frames differ from the one-at-a-time frames %s" % (rep, len(bad), bad[:3]), flush=True)
pools = [torch.cuda.graph_pool_handle() for _ in streams]
gs = {(s, k): Hn.GraphedInterpolator(model, args, frames[k], t, pyramid=pyrs[k], stream=streams[s], pool=pools[s], check=True) for s in range(NS) for k in range(NP)}
torch.cuda.synchronize()
for rep in range(REPS):
    firsts = []
    for i in range(24):
        sk = (i % NS, i % NP)
        gs[sk].replay()
        with torch.cuda.stream(streams[sk[0]]):
            firsts.append((sk[1], gs[sk].out.clone()))
    torch.cuda.synchronize()
    bad = [(i, k, float((o - refs[k]).abs().max())) for i, (k, o) in enumerate(firsts) if not torch.equal(o, refs[k])]
    bad_total += len(bad)
    print("graph replays, 3 streams, rep %d: %d of 24 frames differ %s" % (rep, len(bad), bad[:3]), flush=True)
print("TOTAL differing frames:", bad_total)
sys.exit(1 if bad_total else 0)
