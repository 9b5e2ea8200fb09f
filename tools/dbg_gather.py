"""Feature splats of one pyramid level (both directions): gather kernel vs strip scatter + finish, us per level."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0"); torch.manual_seed(0)
def timeit(fn, n=20):
    for i in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (h, w) in [(288, 480), (288, 512), (144, 240), (72, 120), (36, 60), (18, 30)]:
    feat = torch.rand(1, 96, h, w, device=dev) * 2 - 1
    for amp in (1.0, 4.0, 12.0):
        lo = (torch.rand(1, 4, max(h // 8, 2), max(w // 8, 2), device=dev) - 0.5) * amp + torch.tensor([amp, -amp / 2, -amp, amp / 3], device=dev).view(1, 4, 1, 1)
        up = F.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)
        f1, f0 = feat[:, 48:], feat[:, :48]
        tg = timeit(lambda: hip.softsplat_gather([f1, f0], [up[:, :2], up[:, 2:]], None, "softmax"))
        ts = timeit(lambda: (hip.softsplat_fused(f1, up[:, :2], None, "softmax", want_spk=True), hip.softsplat_fused(f0, up[:, 2:], None, "softmax", want_spk=True)))
        print("%3dx%3d flow ~%4.1f px: gather %.1f us, strip+finish %.1f us" % (h, w, amp, tg, ts), flush=True)
