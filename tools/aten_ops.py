"""Which torch (aten) operators a steady-state 4K forward still launches: torch.profiler table of one warm forward, to find
stray copies / fills between the library calls."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
frames = Hn.frames_from_uint8(Hn.synthetic_pair(int(os.environ.get('FH', 2160)), int(os.environ.get('FW', 3840)), seed=0)).to(dev)
t = torch.tensor([[0.5]], device=dev)
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)
    for _ in range(2):
        Hn.interpolate(model, args, frames, t, pyramid=pyr)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        Hn.interpolate(model, args, frames, t, pyramid=pyr)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=70))
print("==== by source line")
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=50, max_src_column_width=110))
