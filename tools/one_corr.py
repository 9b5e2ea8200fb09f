"""REPS launches of the stand-alone cost-volume operator (fldr_correlation_fwd) at the PWC-Net decoder shapes of a 4K pair
(2176x3840 /64-aligned, both directions batched: N=2; OpticalFlow/PWCNet.py:188,198) for rocprofv3 passes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
shapes = [(2, 196, 34, 60), (2, 128, 68, 120), (2, 96, 136, 240), (2, 64, 272, 480), (2, 32, 544, 960)]
for (n, c, h, w) in shapes:
    a = torch.randn(n, c, h, w, device=dev)
    b = torch.randn(n, c, h, w, device=dev)
    for _ in range(int(os.environ.get("REPS", 5))):
        y = hip.correlation_fwd(a, b)
torch.cuda.synchronize()
