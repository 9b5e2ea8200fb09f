"""Completion time of every step of bench.py's timed region (20 steps, three pairs in flight, graph replay): where the region's
transient is (its ms_per_step is 2-4 % above the sustained loop's).  Prints the completion time of each step after the region's start
and the differences between consecutive completions."""
import os, sys, time, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
NS, NP, K, WU = int(os.environ.get("NS", 3)), 4, int(os.environ.get("K", 20)), 5
t = torch.tensor([[0.5]], device=dev)
frames = [Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=p)).to(dev) for p in range(NP)]
with torch.no_grad():
    pyrs = [Hn.build_pyramid(Hn.pad_frames(f, args), args) for f in frames]
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
pools = [torch.cuda.graph_pool_handle() for _ in streams]
graphs = {(s, k): Hn.GraphedInterpolator(model, args, frames[k], t, pyramid=pyrs[k], stream=streams[s], pool=pools[s], check=False) for s in range(NS) for k in range(NP)}
torch.cuda.synchronize()
def step(i): return graphs[(i % NS, i % NP)].replay()
# calibrate torch.cuda._sleep (cycles of some device counter) -> microseconds; also loads its module outside every timed region
ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
ea.record(); torch.cuda._sleep(1000000); eb.record(); torch.cuda.synchronize()
CYC_PER_US = 1000000 / (ea.elapsed_time(eb) * 1e3)
print("torch.cuda._sleep: %.1f cycles per us" % CYC_PER_US)
STAGGER = [float(x) for x in os.environ.get("STAGGER_US", "0").split(",")]
for rep in range(3 * len(STAGGER)):
    stag = STAGGER[rep % len(STAGGER)]
    for i in range(WU): step(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream())
    for s in streams: s.wait_event(e0)
    t0 = time.perf_counter()
    if stag > 0:
        for si in range(1, NS):
            with torch.cuda.stream(streams[si]):
                torch.cuda._sleep(int(si * stag * CYC_PER_US))
    evs = []
    for i in range(K):
        step(i)
        e = torch.cuda.Event(enable_timing=True); e.record(streams[i % NS]); evs.append(e)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done = sorted(e0.elapsed_time(e) for e in evs)
    print("stagger %g us, rep %d: %d steps in %.2f ms (%.3f per step; host enqueue %.2f ms); completions (ms): %s" % (stag, rep, K, dt * 1e3, dt * 1e3 / K, t_host * 1e3, " ".join("%.2f" % x for x in done)))
    print("        gaps: %s" % " ".join("%.2f" % (b - a) for a, b in zip([0.0] + done, done)))
