"""NF single-stream 4K forwards (pyramid prebuilt) for a rocprofv3 kernel trace; tools/trace_timeline.py reads the CSV."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn, fldr_hip
if os.environ.get("LIB"): fldr_hip.LIB_PATH = os.environ["LIB"]      # experimental build (tools/stamps/build_variant.sh)
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
frames = Hn.frames_from_uint8(Hn.synthetic_pair(int(os.environ.get('FH', 2160)), int(os.environ.get('FW', 3840)), seed=0)).to(dev)
t = torch.tensor([[0.5]], device=dev)
with torch.no_grad():
    pyr = Hn.build_pyramid(Hn.pad_frames(frames, args), args)
    for _ in range(int(os.environ.get("NF", 4))):
        Hn.interpolate(model, args, frames, t, pyramid=pyr)
        torch.cuda.synchronize()
