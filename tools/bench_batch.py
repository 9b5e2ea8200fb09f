"""Throughput with B pairs per forward (batch dimension) x S streams: ms per PAIR.  The reference's test loop runs B=1."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
with torch.no_grad():
    for B, S in ((1, 3), (2, 2), (2, 3), (3, 2), (4, 1), (4, 2)):
        frames = torch.cat([Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=i)) for i in range(B)], 0).to(dev)
        t = torch.full((B, 1), 0.5, device=dev)
        pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)
        streams = [torch.cuda.Stream() for _ in range(S)]
        for s in streams: s.wait_stream(torch.cuda.current_stream())
        def step(i):
            with torch.cuda.stream(streams[i % S]):
                return Hn.interpolate(model, args, frames, t, pyramid=pyr)
        for i in range(2 * S): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 12
        for i in range(n): o = step(i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("B=%d streams=%d: %.3f ms per pair (%.1f pairs/s)" % (B, S, dt / (n * B) * 1e3, n * B / dt), flush=True)
        del frames, pyr, o
        torch.cuda.empty_cache()
