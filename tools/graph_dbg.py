import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_harness as Hn, fldr_hip as hip
dev = torch.device("cuda:0")
model, _, args = Hn.prepare_model(dev)
t = torch.tensor([[0.5]], device=dev)
H, W = int(os.environ.get("H", 2160)), int(os.environ.get("W", 3840))
share = os.environ.get("SHARE", "1") == "1"
with torch.no_grad():
    prs = []
    for k in range(3):
        fr = Hn.frames_from_uint8(Hn.synthetic_pair(H, W, seed=k)).to(dev)
        prs.append((fr, Hn.build_pyramid(Hn.pad_frames(fr, args), args)))
    st = torch.cuda.Stream()
    refs = []
    with torch.cuda.stream(st):
        for k in range(3):
            for i in range(2): r = Hn.interpolate(model, args, prs[k][0], t, pyramid=prs[k][1])
            refs.append(r.clone())
    torch.cuda.synchronize()
    pool = torch.cuda.graph_pool_handle()
    gs, outs = [], []
    for k in range(3):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool if share else None, stream=st):
            outs.append(Hn.interpolate(model, args, prs[k][0], t, pyramid=prs[k][1]))
        gs.append(g)
    torch.cuda.synchronize()
    for order in ([0, 1, 2], [2, 1, 0], [0, 0, 1], [1, 2, 0]):
        res = []
        for k in order:
            with torch.cuda.stream(st): gs[k].replay()
            torch.cuda.synchronize()
            res.append("%d:%s" % (k, "ok" if torch.equal(outs[k], refs[k]) else "DIFF %.2e" % (outs[k] - refs[k]).abs().max().item()))
        print("share" if share else "own pools", order, res)
    # back to back without syncs in between
    with torch.cuda.stream(st):
        for k in (0, 1, 2, 0, 1, 2): gs[k].replay()
    torch.cuda.synchronize()
    print("after back-to-back: last of each:", ["ok" if torch.equal(outs[k], refs[k]) else "DIFF" for k in range(3)])
