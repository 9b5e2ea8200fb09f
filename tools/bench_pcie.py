"""PCIe-inclusive rate (DESIGN.md note, never the bench `value`): uint8 frame pair in pinned host memory -> H2D -> ingest kernels ->
forward -> rounded uint8 frame -> D2H into pinned host memory, per pair, single stream and 3 streams."""
import os, sys, time, torch
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "fldr-vfi_amd"))
import fldr_harness as Hn
import fldr_hip
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
H, W = 2160, 3840
u8_host = Hn.synthetic_pair(H, W, seed=0).unsqueeze(0).pin_memory()
out_host = torch.empty(1, 3, H, W, dtype=torch.uint8).pin_memory()
t = torch.tensor([[0.5]], device=dev)
def one(stream):
    with torch.cuda.stream(stream):
        u8 = u8_host.to(dev, non_blocking=True)
        img, _ = Hn.interpolate_u8(model, args, u8, t)          # fused ingest + pyramid, forward, the rounded 8-bit frame straight from the synthesis kernel
        out_host.copy_(img, non_blocking=True)
with torch.no_grad():
    for ns in (1, 3):
        streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        for i in range(2 * ns): one(streams[i % ns])
        torch.cuda.synchronize()
        n = 12
        t0 = time.perf_counter()
        for i in range(n): one(streams[i % ns])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%d stream(s): %.2f ms per pair incl. H2D of 49.8 MB uint8 frames and D2H of the 24.9 MB uint8 result = %.1f pairs/s" % (ns, dt * 1e3, 1 / dt))
