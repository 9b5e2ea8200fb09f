"""dec3 + softmax/T + blend at the 4K shape: tile-grid shift 0 / 16 (us per launch, rotating inputs)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0"); torch.manual_seed(0); L = hip.lib()
H, W = 2304, int(os.environ.get("FW", 3840))
sets = [(torch.rand(1, 16, H // 2, W // 2, device=dev), [torch.rand(1, 3, H, W, device=dev) * 2 - 1 for _ in range(6)]) for _ in range(2)]
wt = torch.randn(6, 16, 3, 3, device=dev) / 6; bs = torch.randn(6, device=dev) * 0.3; t = torch.tensor([[0.5]], device=dev)
def timeit(fn, n=16):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for xs in (0, 16, 0, 16, 8):
    L.fldr_debug_dec3_xshift(xs)
    print("x shift %2d: %.1f us" % (xs, timeit(lambda i: hip.dec3_synth(sets[i % 2][0], wt, bs, sets[i % 2][1], t, 1.5616))), flush=True)
L.fldr_debug_dec3_xshift(-1)
