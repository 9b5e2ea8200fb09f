"""Coarse pyramid levels of the 96->96 3x3 convolution: 48-channel groups (fldr_debug_spk_small_units(-1)) against
16-channel sub-groups of the same weight pack; prints us per launch."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
wt = torch.randn(96, 96, 3, 3, device=dev) / 30
b = torch.randn(96, device=dev)
for (h, w) in ((9, 15), (18, 30), (36, 60), (72, 120), (144, 240), (288, 480)):
    x = torch.rand(1, 96, h, w, device=dev) * 2 - 1
    xp = hip.spk_pack(x)
    row = []
    ref = None
    for su in (-1, 96, 160, 320, 100000):
        hip.lib().fldr_debug_spk_small_units(su)
        for _ in range(5):
            y = hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = hip.conv2d_spk([xp], wt, b, relu=True, want_f32=False, want_spk=True)
        e1.record(); torch.cuda.synchronize()
        row.append("%6.1f" % (e0.elapsed_time(e1) * 1000 / 50))
        if ref is None: ref = y.buf.clone()
        assert torch.equal(ref, y.buf)
    print(h, w, "small_units -1/96/160/320/all:", " ".join(row), flush=True)
hip.lib().fldr_debug_spk_small_units(96)
