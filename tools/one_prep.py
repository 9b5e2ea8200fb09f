#!/usr/bin/env python3
"""level0_prep at the 4K shape a few times (for rocprofv3 --pmc passes): one_prep.py [phase]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import torch
import fldr_hip as hip
dev = torch.device("cuda:0")
H, W, up = 2304, 3840, 8
phase = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lo = torch.tensor([-0.75, -0.5, 0.75, 0.5], device=dev).view(1, 4, 1, 1) + torch.randn(1, 4, H // up, W // up, device=dev) * 0.02
frames = [torch.rand(1, 3, 2, H, W, device=dev) * 2 - 1 for _ in range(3)]
t4 = torch.tensor([0.5], device=dev).view(1, 1, 1, 1)
for i in range(6):
    hip.level0_prep(lo, frames[i % 3][:, :, 0], frames[i % 3][:, :, 1], t4, H, W, 20.0, 20.0, withmask=True, want_z=True)
torch.cuda.synchronize()
