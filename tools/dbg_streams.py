import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
frames = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=0)).to(dev)
t = torch.tensor([[0.5]], device=dev)
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)
    for ns in (1, 2, 3, 4, 5, 6):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        for s in streams: s.wait_stream(torch.cuda.current_stream())
        def run(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % ns]):
                    out = Hn.interpolate(model, args, frames, t, pyramid=pyr)
            return out
        run(4); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(24); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("streams", ns, "ms/step %.3f" % (dt / 24 * 1e3))
