#!/usr/bin/env python3
"""dec2 + dec3 at the 4K shape: the two kernels (conv2d_spk 48 -> 16 on nearest-x2(dec1) + enc1, then dec3_synth on the packed tensor)
against the fused producer / consumer kernel (dec23_synth), rotating inputs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import torch
import fldr_hip as hip
dev = torch.device("cuda:0")
H, W = 2304, 3840
h, w = H // 2, W // 2
torch.manual_seed(0)
sets = []
for k in range(3):
    dec1p = hip.spk_pack(torch.rand(1, 32, h // 2, w // 2, device=dev) * 1.5)
    enc1p = hip.spk_pack(torch.rand(1, 16, h, w, device=dev) * 1.5)
    cands = [torch.rand(1, 3, H, W, device=dev) * 2 - 1 for _ in range(6)]
    sets.append((dec1p, enc1p, cands))
w2 = torch.randn(16, 48, 3, 3, device=dev) / 12
b2 = torch.randn(16, device=dev) * 0.2
w3 = torch.randn(6, 16, 3, 3, device=dev) / 6
b3 = torch.randn(6, device=dev) * 0.3
t = torch.tensor([[0.5]], device=dev)
def two(i):
    d1, e1, c = sets[i % 3]
    d2p = hip.conv2d_spk([d1, e1], w2, b2, relu=True, up2=[True, False], want_f32=False, want_spk=True)
    return hip.dec3_synth(d2p, w3, b3, c, t, 1.5616)
def fused(i):
    d1, e1, c = sets[i % 3]
    return hip.dec23_synth(d1, e1, w2, b2, w3, b3, c, t, 1.5616)
def timeit(fn, n=18):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    print("dec2 + dec3 (two kernels): %.1f us   fused: %.1f us   max |diff| %.2e" % (timeit(two), timeit(fused), (two(0) - fused(0)).abs().max().item()), flush=True)
if hasattr(hip.lib(), "fldr_debug_read_d23_stamps"):
    import ctypes
    fused(0); torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 16)()
    hip.lib().fldr_debug_read_d23_stamps.argtypes = [ctypes.c_void_p]
    hip.lib().fldr_debug_read_d23_stamps(buf)
    n = max(1, buf[3])
    print("consumer wave 0: tiles %d | per tile (s_memtime ticks): cands+matrix+exchange %d (cands issue %d)  softmax %d  dma-issue %d  blend+store %d  barrier %d" % (buf[3], buf[0] / n, buf[6] / n, buf[5] / n, buf[4] / n, buf[1] / n, buf[2] / n))
    n = max(1, buf[13])
    print("producer wave 8: tiles %d | per tile: stage-issue %d  mfma-loop %d  epilogue %d  dma-wait %d  barrier %d" % (buf[13], buf[8] / n, buf[9] / n, buf[10] / n, buf[11] / n, buf[12] / n))
if hasattr(hip.lib(), "fldr_debug_read_d23_wave_stamps"):
    wb = (ctypes.c_uint64 * 24)()
    hip.lib().fldr_debug_read_d23_wave_stamps.argtypes = [ctypes.c_void_p]
    hip.lib().fldr_debug_read_d23_wave_stamps(wb)
    nt = max(1, buf[3])
    print("per wave and tile: barrier wait / loop cycles  " + "  ".join("w%d %d/%d" % (w, wb[2 * w] / nt, wb[2 * w + 1] / nt) for w in range(12)))
