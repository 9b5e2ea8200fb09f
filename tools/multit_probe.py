import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn, fldr_hip
dev = torch.device("cuda:0")
m, _, a = Hn.prepare_model(dev)
frames = Hn.frames_from_uint8(Hn.synthetic_pair(256, 384, seed=3, quadrant=True)).to(dev)
ts = [k / 8 for k in range(1, 8)]
for rep in range(int(os.environ.get('REPS', 3))):
    cached = Hn.interpolate_multi(m, a, frames, ts)
    for tv, c in zip(ts, cached):
        t = torch.tensor([[tv]], device=dev)
        p1 = Hn.interpolate(m, a, frames, t)
        p2 = Hn.interpolate(m, a, frames, t)
        d = (c - p1).abs(); d2 = (p1 - p2).abs()
        if os.environ.get("QUIET") and max(d.max().item(), d2.max().item()) < 1e-4: continue
        print(os.environ.get("FLDR_SPLAT_BOUNDS", "lowres"), rep, tv, "cached-plain max %.2e frac>2e-5 %.2e | plain-plain max %.2e frac %.2e" % (d.max().item(), (d > 2e-5).float().mean().item(), d2.max().item(), (d2 > 2e-5).float().mean().item()))
