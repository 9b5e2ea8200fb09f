"""8x interpolation (t = 1/8..7/8) of one 4096x2160 pair (BASELINE config 3 geometry): with and without the pair-invariant cache."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
frames = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 4096, seed=0)).to(dev)
ts = [k / 8 for k in range(1, 8)]
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)
    for name, fn in (("uncached", lambda: [Hn.interpolate(model, args, frames, torch.tensor([[tv]], device=dev), pyramid=pyr) for tv in ts]),
                     ("pair cache", lambda: Hn.interpolate_multi(model, args, frames, ts, pyramid=pyr))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("%-10s: %.2f ms per pair (7 outputs) = %.1f output frames/s" % (name, dt * 1e3, 7 / dt))
