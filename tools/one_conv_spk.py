"""REPS launches of the dominant kernel (conv3x3_ring_kernel<3,3,false,8,32>, 96->96 3x3 @288x480, packed in / packed out) for the
rocprofv3 PMC passes (profiles/r0N_conv96_*)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
h, w = int(os.environ.get("CH", 288)), int(os.environ.get("CW", 480))
x = torch.rand(1, 96, h, w, device=dev) * 2 - 1
xp = hip.spk_pack(x)
wt = torch.randn(96, 96, 3, 3, device=dev) / 30
b = torch.randn(96, device=dev)
for _ in range(int(os.environ.get("REPS", 10))):
    y = hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True)
torch.cuda.synchronize()
