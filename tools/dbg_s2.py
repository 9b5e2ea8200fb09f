"""Stride-2 split conv vs the exact fp32-MFMA kernel and an fp64 reference; timing on the three encoder shapes."""
import os, sys, torch
import torch.nn.functional as F
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (parts, cout, H, W, N) in [([3, 3, 2, 5], 16, 40, 72, 1), ([16], 32, 34, 70, 2), ([32], 64, 48, 64, 1), ([26], 16, 50, 38, 1),
                               ([3, 3, 3, 3, 2, 2, 2, 2, 3, 3], 16, 2304, 3840, 1), ([16], 32, 1152, 1920, 1), ([32], 64, 576, 960, 1)]:
    srcs = [torch.randn(N, c, H, W, device=dev) for c in parts]
    cin = sum(parts)
    wt = torch.randn(cout, cin, 4, 4, device=dev) / (cin * 16) ** 0.5
    b = torch.randn(cout, device=dev)
    a32 = hip.conv2d(srcs, wt, b, stride=2, relu=True, precision="fp32")
    sp, spk = hip.conv2d(srcs, wt, b, stride=2, relu=True, precision="split", want_spk=True)
    torch.cuda.synchronize()
    msg = ""
    if H * W < 1e5:
        ref = F.relu(F.conv2d(torch.cat(srcs, 1).double(), wt.double(), b.double(), stride=2, padding=1))
        msg = "| vs fp64: fp32-MFMA max %.2e mean %.2e, split max %.2e mean %.2e" % ((a32 - ref).abs().max().item(), (a32 - ref).abs().mean().item(),
                                                                                (sp - ref).abs().max().item(), (sp - ref).abs().mean().item())
    pk_ok = torch.equal(hip.spk_pack(sp).buf, spk.buf)
    print("src%s cout %d %dx%d N%d: split vs fp32-MFMA max diff %.2e, packed twin ok %s %s | fp32 %.1f us, split %.1f us" % (
        parts, cout, H, W, N, (a32 - sp).abs().max().item(), pk_ok, msg,
        timeit(lambda: hip.conv2d(srcs, wt, b, stride=2, relu=True, precision="fp32", want_spk=True)),
        timeit(lambda: hip.conv2d(srcs, wt, b, stride=2, relu=True, precision="split", want_spk=True))))
