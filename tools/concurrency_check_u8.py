"""The uint8-in -> uint8-out unit of work (fused ingest + pyramid, forward, rounded 8-bit frame from the synthesis kernel) with DIFFERENT pairs in
flight on three HIP streams against the same calls one at a time: the same bytes.   python tools/concurrency_check_u8.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
t = torch.tensor([[0.5]], device=dev)
NS, NP = 3, 4
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
u8 = [Hn.synthetic_pair(2160, 3840, seed=60 + p).unsqueeze(0).to(dev) for p in range(NP)]
refs = [Hn.interpolate_u8(model, args, u8[k], t)[0].clone() for k in range(NP)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
bad_total = 0
for rep in range(REPS):
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    outs = []
    for i in range(12):
        with torch.cuda.stream(streams[i % NS]):
            outs.append((i % NP, Hn.interpolate_u8(model, args, u8[i % NP], t)[0]))
    torch.cuda.synchronize()
    bad = [(i, k, int((o != refs[k]).sum())) for i, (k, o) in enumerate(outs) if not torch.equal(o, refs[k])]
    bad_total += len(bad)
    print("uint8 path, 3 streams, rep %d: %d of 12 frames differ %s" % (rep, len(bad), bad[:3]), flush=True)
print("TOTAL differing frames:", bad_total)
sys.exit(1 if bad_total else 0)
