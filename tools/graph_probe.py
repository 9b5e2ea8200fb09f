#!/usr/bin/env python3
"""The bench loop (3 streams, 4 rotated 4K pairs) eager vs hipGraph replay of whole forwards (one graph per (stream, pair), a memory pool
per stream): does replay help when the host is the slow side?  Also checks that the replayed frame has the bits of the eager one."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
t = torch.tensor([[0.5]], device=dev)
NS, NP = int(os.environ.get("STREAMS", 3)), 4
pairs = []
with torch.no_grad():
    for k in range(NP):
        fr = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=k)).to(dev)
        pairs.append((fr, Hn.build_pyramid(Hn.pad_frames(fr, args), args)))
streams = [torch.cuda.Stream() for _ in range(NS)]
def eager(i):
    with torch.cuda.stream(streams[i % NS]), torch.no_grad():
        fr, pyr = pairs[i % NP]
        return Hn.interpolate(model, args, fr, t, pyramid=pyr)
for i in range(12): ref = eager(i)
torch.cuda.synchronize()
refs = []
for k in range(NP):
    with torch.cuda.stream(streams[k % NS]):
        refs.append(eager(k).clone())            # (cloned on the stream that produced it)
torch.cuda.synchronize()
def loop(fn, n):
    for i in range(12): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
e = loop(eager, 120)
print("eager : host enqueue %.3f ms/forward, total %.3f ms/forward" % e, flush=True)
graphs, outs, pools = {}, {}, [torch.cuda.graph_pool_handle() for _ in range(NS)]
t0 = time.perf_counter()
for s in range(NS):
    for k in range(NP):
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g, pool=pools[s], stream=streams[s]):
            fr, pyr = pairs[k]
            outs[(s, k)] = Hn.interpolate(model, args, fr, t, pyramid=pyr)
        graphs[(s, k)] = g
torch.cuda.synchronize()
print("captured %d graphs in %.1f s; memory allocated %.1f GB, reserved %.1f GB" % (len(graphs), time.perf_counter() - t0, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9), flush=True)
def check(tag):
    bad = []
    for s in range(NS):
        for k in range(NP):
            with torch.cuda.stream(streams[s]): graphs[(s, k)].replay()
            torch.cuda.synchronize()
            if not torch.equal(outs[(s, k)], refs[k]): bad.append((s, k, float((outs[(s, k)] - refs[k]).abs().max())))
    print(tag, "mismatching (stream, pair, max diff):", bad if bad else "none", flush=True)
check("right after capture:")
def replay(i):
    s, k = i % NS, i % NP
    with torch.cuda.stream(streams[s]):
        graphs[(s, k)].replay()
    return outs[(s, k)]
r = loop(replay, 120)
print("graphs: host enqueue %.3f ms/forward, total %.3f ms/forward" % r, flush=True)
check("after the concurrent replay loop:")

e = loop(eager, 120); r = loop(replay, 120)
print("eager : total %.3f ms/forward | graphs: total %.3f ms/forward (host %.3f)" % (e[1], r[1], r[0]))
