"""Does bench.py's 20-step timed region depend on how long the GPU was loaded right before it?  Each trial: 1 s of idle (the state after
graph capture), P untimed steps, synchronise, the 5 warm-up steps, synchronise, then 20 timed steps (three pairs in flight, graph replay)."""
import os, sys, time, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
NS, NP, K, WU = 3, 4, 20, 5
t = torch.tensor([[0.5]], device=dev)
frames = [Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=p)).to(dev) for p in range(NP)]
with torch.no_grad():
    pyrs = [Hn.build_pyramid(Hn.pad_frames(f, args), args) for f in frames]
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
pools = [torch.cuda.graph_pool_handle() for _ in streams]
graphs = {(s, k): Hn.GraphedInterpolator(model, args, frames[k], t, pyramid=pyrs[k], stream=streams[s], pool=pools[s], check=True) for s in range(NS) for k in range(NP)}
torch.cuda.synchronize()
def step(i): return graphs[(i % NS, i % NP)].replay()
for trial in range(int(os.environ.get("TRIALS", 3))):
    for P in [int(x) for x in os.environ.get("PREROLL", "0,12,48").split(",")]:
        time.sleep(float(os.environ.get("IDLE_S", 1.0)))
        for i in range(P): step(i)
        torch.cuda.synchronize()
        for i in range(WU): step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K): step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("pre-roll %3d steps: region %.2f ms = %.3f ms per step" % (P, dt * 1e3, dt * 1e3 / K), flush=True)
