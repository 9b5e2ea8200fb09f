"""How long the host takes to ENQUEUE one forward (Python + ctypes + allocator) against the GPU time per pair: if the two are
close, the three-stream bench is host-bound and kernel work alone cannot raise it."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
t = torch.tensor([[0.5]], device=dev)
pairs = []
with torch.no_grad():
    for k in range(4):
        fr = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=k)).to(dev)
        pairs.append((fr, Hn.build_pyramid(Hn.pad_frames(fr, args), args)))
    streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("STREAMS", 3)))]
    def step(i):
        with torch.cuda.stream(streams[i % len(streams)]):
            fr, pyr = pairs[i % 4]
            return Hn.interpolate(model, args, fr, t, pyramid=pyr)
    for i in range(12): step(i)
    torch.cuda.synchronize()
    for n in (30, 60):
        t0 = time.perf_counter()
        for i in range(n): step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%d forwards on %d streams: enqueue %.2f ms/forward, total %.2f ms/forward (host waits %.1f ms at the end)" % (n, len(streams), (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t2 - t1) * 1e3))
    # host cost alone: a stream whose work is already far behind (enqueue again without draining)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(12): step(i)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(14)
