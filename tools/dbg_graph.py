import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
frames = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=0)).to(dev)
t = torch.tensor([[0.5]], device=dev)
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)
    ref = Hn.interpolate(model, args, frames, t, pyramid=pyr).clone()
    for ns in (3, 4, 5):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        graphs, outs = [], []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2): Hn.interpolate(model, args, frames, t, pyramid=pyr)      # warm-up on this stream
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                o = Hn.interpolate(model, args, frames, t, pyramid=pyr)
            graphs.append(g); outs.append(o)
        torch.cuda.synchronize()
        def run(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % ns]):
                    graphs[i % ns].replay()
        run(2 * ns); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(24); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        err = (outs[0] - ref).abs().max().item()
        print("graph streams", ns, "ms/step %.3f" % (dt / 24 * 1e3), "max|graph-eager| %.2e" % err)
        del graphs, outs
