"""Which torch operators of one warm 4K forward launch copies / fills (the __amd_rocclr_copyBuffer and aten kernels of the rocprof table)?
torch.profiler with stacks; prints every device-side memcpy / memset / aten kernel with the Python frame that issued it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fldr-vfi_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import fldr_harness as Hn

dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
f = Hn.frames_from_uint8(Hn.synthetic_pair(2160, 3840, seed=0)).to(dev)
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(f, args), args)
    t = torch.tensor([[0.5]], device=dev)
    for _ in range(3):
        Hn.interpolate(model, args, f, t, pyramid=pyr)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        Hn.interpolate(model, args, f, t, pyramid=pyr)
        torch.cuda.synchronize()
rows = []
for ev in prof.events():
    n = ev.name
    if n.startswith("aten::") and ev.cpu_parent is None or (ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::") and n.startswith("aten::")):
        st = [s for s in (ev.stack or []) if "fldr-vfi_amd" in s]
        rows.append((n, ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total, st[0] if st else (ev.stack[0] if ev.stack else "?")))
from collections import Counter
c = Counter((n, s) for n, _, s in rows)
for (n, s), k in sorted(c.items(), key=lambda x: -x[1]):
    print("%3d x %-28s %s" % (k, n, s))
