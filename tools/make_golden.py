#!/usr/bin/env python3
"""Generate tests/golden/*.npz and the plain-tensor weight archive by importing
the reference's own model code on CPU.  Runs ONLY in the build container (needs
/root/reference); nothing here travels to the GPU box except its outputs.

    python -O tools/make_golden.py          # -O: fLDRnet.py:448 asserts get_device()==gpu

Recipe = SURVEY.md Appendix D: the third-party packages the reference imports
but this image lacks (cupy, cv2, skimage, torchvision) are registered as inert
placeholders (none is touched by the test-path model code), the checkpoint is
loaded with weights_only=True and an allow-list, and the one operator with no
CPU implementation in the reference (the CUDA softmax splat) is supplied by
oracle/fldr_oracle.py.  Everything else on the path is the reference's code.
"""
import os
import sys
import types
from collections import Counter, OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fldr_oracle as O  # noqa: E402


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Inert(self.__name__ + "." + name)
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return _Inert("call")


def import_reference():
    for name in ("cupy", "cv2", "skimage", "skimage.feature", "skimage.metrics", "skimage.transform",
                 "torchvision", "torchvision.transforms", "torchvision.models", "torchvision.utils"):
        sys.modules[name] = _Inert(name)
    sys.modules["cupy"].memoize = lambda **k: (lambda f: f)
    real = torch.cuda.current_stream
    torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=0)  # correlation.py:7-8
    sys.path.insert(0, REF)
    argv, sys.argv = sys.argv, ["x"]
    try:
        import run_on_your_images as R
        import fLDRnet
        import pca_comp
    finally:
        torch.cuda.current_stream = real
    args = R.args_config()
    sys.argv = argv
    args.gpu = "cpu"
    return R, fLDRnet, pca_comp, args


def load_model(fLDRnet, pca_comp, args):
    import numpy
    ck = os.path.join(REF, "checkpoint_dir/fLDRnet_X4K1000FPS_exp1/fLDRnet_X4K1000FPS_exp1_best_PSNR.pt")
    import _codecs
    safe = [pca_comp.DCTParams, Counter, OrderedDict, numpy.dtype, _codecs.encode,
            (numpy._core.multiarray.scalar, "numpy.core.multiarray.scalar")]
    safe += [type(numpy.dtype(t)) for t in ("float64", "float32", "int64", "int32")]
    torch.serialization.add_safe_globals(safe)
    ckpt = torch.load(ck, map_location="cpu", weights_only=True)
    model = fLDRnet.DCTXVFInet(args)
    print(model.load_state_dict(ckpt["state_dict_Model"]))
    model.save_params([pca_comp.DCTParams(8, 0.25, 0.5) for _ in range(6)])
    model.eval()

    class CpuSplat(torch.nn.Module):       # the only non-reference code on the path
        def forward(self, img, flow, z=None):
            return O.function_softsplat(img, flow, z, "softmax")

    model.vfinet.softsplat = CpuSplat()
    return model, ckpt


def export_weights(ckpt, path):
    sd = ckpt["state_dict_Model"]
    keep = {}
    for k, v in sd.items():
        if k.startswith("base_modules."):
            continue                      # aliases of rec_ctx_ds.* / vfinet.* 
        if k in ("EV16", "EV32", "EV64", "Mean16", "Mean32", "Mean64", "meanVec16", "meanVec32", "meanVec64"):
            continue                      # unused at inference (index8 = 0 everywhere, fLDRnet.py:135)
        if ".refine_unet.conv1." in k or ".refine_unet.conv2." in k:
            continue                      # never called (fLDRnet.py:589-590, 619-644)
        keep[k] = v.detach().cpu().numpy()
    np.savez(path, **keep)
    meta = {k: (ckpt[k] if not torch.is_tensor(ckpt[k]) else ckpt[k].item())
            for k in ("net_type", "last_epoch", "best_PSNR", "testPSNR")}
    print("exported", len(keep), "tensors", sum(v.size for v in keep.values()), "elements", meta)
    return {k: torch.from_numpy(v) for k, v in keep.items()}


def crops_of(t, crops):
    return np.stack([t[..., y0:y0 + h, x0:x0 + w].numpy() for (y0, x0, h, w) in crops], 0)


def model_case(model, R, fLDRnet, args, H, W, tval, seed, quadrant, out_path, slim=False):
    u8 = O.synthetic_pair(H, W, seed=seed, quadrant=quadrant)
    frames = O.frames_from_uint8(u8)
    # caller code of the reference, run_on_your_images.py:117-153
    B, C, T, _, _ = frames.shape
    x = frames.reshape(B, -1, H, W)
    div = (2 ** args.S_tst) * 8
    ph, pw = (div - H % div) % div, (div - W % div) % div
    x = torch.nn.functional.pad(x, (0, pw, 0, ph), args.padding).reshape(B, C, T, H + ph, W + pw)
    Bp, Cp, Tp, Hp, Wp = x.shape
    pyr = [torch.nn.functional.interpolate(
        x.permute(0, 2, 1, 3, 4).reshape(Bp * Tp, Cp, Hp, Wp), scale_factor=args.scales[0] / args.scales[i],
        mode="bicubic", align_corners=args.align_cornerse).reshape(
            Bp, Tp, Cp, int(Hp * (args.scales[0] / args.scales[i])), int(Wp * (args.scales[0] / args.scales[i]))
        ).permute(0, 2, 1, 3, 4) if i != 0 else x for i in range(args.S_tst + 1)]
    t = torch.tensor([[tval]], dtype=torch.float32)

    rec = {}
    # hooks on the reference's own modules
    import pca_comp
    pcas = []
    orig_pca = fLDRnet.to_pca_diff

    def pca_hook(*a, **k):
        r = orig_pca(*a, **k)
        pcas.append(r.clone())
        return r
    fLDRnet.to_pca_diff = pca_hook
    feats = []
    h1 = model.rec_ctx_ds.register_forward_hook(lambda m, i, o: feats.append((o + i[0]).clone()))
    lv = []

    orig_vfi_forward = model.vfinet.forward

    def vfi_hook(*a, **k):
        r = orig_vfi_forward(*a, **k)
        lv.append((k["level"], r))
        return r
    model.vfinet.forward = vfi_hook
    unet_io = []
    h2 = model.vfinet.refine_unet.register_forward_hook(lambda m, i, o: unet_io.append((i[0].clone(), o.clone())))
    bw = []
    orig_bwarp = model.vfinet.bwarp

    def bwarp_hook(*a, **k):
        r = orig_bwarp(*a, **k)
        bw.append(r.clone())
        return r
    model.vfinet.bwarp = bwarp_hook

    with torch.no_grad():
        inp_list = [torch.zeros(B, 96, Hp // 8, Wp // 8) for _ in range(6)]
        out, _ = model(inp_list, t, normInput=[p.clone() for p in pyr], is_training=False, validation=False)
    fLDRnet.to_pca_diff = orig_pca
    h1.remove(); h2.remove()
    model.vfinet.forward = orig_vfi_forward
    model.vfinet.bwarp = orig_bwarp
    assert out.dtype == torch.float64, out.dtype

    crops = [(0, 0, 40, 64), (Hp - 40, Wp - 64, 40, 64), (Hp // 2 - 20, Wp // 2 - 32, 40, 64)]
    cat, refine_out = unet_io[0]
    # bwarp call order in DCTVFInet.forward: im_1_0, im_0_1, flowback_0, flowback_1, im0_tot, im1_tot
    rec["frames_u8"] = u8.numpy()
    rec["t"] = np.float32(tval)
    rec["crops"] = np.array(crops, dtype=np.int64)
    for i in range(6):
        if i == 0 and slim:
            continue
        rec["pyr%d" % i] = pyr[i].numpy() if i > 0 else np.zeros(0, np.float32)   # level 0 = padded frames
        rec["pca%d" % i] = pcas[i].reshape(1, 96, Hp // 8 >> i, Wp // 8 >> i).float().numpy()
        rec["feat%d" % i] = feats[i].numpy()
    for level, r in lv:
        if level != 0:
            rec["flow%d" % level] = r.numpy()
    rec["cat26_crops"] = crops_of(cat, crops)               # [ncrop,1,26,48,96]
    rec["refine_out_crops"] = crops_of(refine_out, crops)
    rec["im_1_0_crops"] = crops_of(bw[0], crops)
    rec["im_0_1_crops"] = crops_of(bw[1], crops)
    rec["cat26_sum"] = cat.double().sum((0, 2, 3)).numpy()
    rec["cat26_abssum"] = cat.double().abs().sum((0, 2, 3)).numpy()
    rec["refine_out_sum"] = refine_out.double().sum((0, 2, 3)).numpy()
    rec["out"] = out.float().numpy() if slim else out.numpy()   # fp64 [1,3,Hp,Wp] (cropped to <=2160x4096)
    rec["out_dtype"] = str(out.dtype)
    np.savez_compressed(out_path, **rec)
    print("wrote", out_path, os.path.getsize(out_path) >> 10, "KiB  out", tuple(out.shape))
    return rec


def identity_splat_case(model, args, rec, out_path):
    """(4') per-level flows with the splat replaced by identity: pins the convs and
    resizes independently of the splat restatement."""
    class Ident(torch.nn.Module):
        def forward(self, img, flow, z=None):
            return img
    keep = model.vfinet.softsplat
    model.vfinet.softsplat = Ident()
    lv = {}
    with torch.no_grad():
        flow = None
        for level in range(5, 0, -1):
            feat = torch.from_numpy(rec["feat%d" % level])
            flow = model.vfinet(feat, flow, torch.tensor([[0.5]]).view(1, 1, 1, 1), level=level, is_training=False,
                                normInput=None, validation=False)
            lv["flow%d" % level] = flow.numpy()
    model.vfinet.softsplat = keep
    np.savez_compressed(out_path, **lv)
    print("wrote", out_path)


def op_cases(model, fLDRnet, pca_comp, args, out_path):
    g = torch.Generator().manual_seed(7)
    rec = {}
    # (6) bwarp alone, incl. out-of-range flows
    x = torch.rand(1, 3, 40, 72, generator=g) * 2 - 1
    flo = (torch.rand(1, 2, 40, 72, generator=g) - 0.5) * 30
    flo[:, :, :4, :] *= 10
    with torch.no_grad():
        rec["bwarp_x"], rec["bwarp_flo"] = x.numpy(), flo.numpy()
        rec["bwarp_out"] = model.vfinet.bwarp(x, flo, withmask=True).numpy()
        rec["bwarp_out_nomask"] = model.vfinet.bwarp(x, flo, withmask=False).numpy()
        # (7) refine UNet alone
        u = torch.rand(1, 26, 64, 96, generator=g) * 2 - 1
        rec["unet_in"] = u.numpy()
        rec["unet_out"] = model.vfinet.refine_unet(u).numpy()
        # (2) to_pca_diff alone on random planes
        pl = torch.rand(6, 32, 48, generator=g) * 2 - 1
        rec["pca_in"] = pl.numpy()
        rec["pca_out"] = pca_comp.to_pca_diff(pl, model.params[0], args, model.pca_means[0], model.EVs[0],
                                              model.mean_vecs[0]).numpy()
        # rec_ctx_ds / conv_flow_bottom / conv_flow1 / conv_flow2 alone
        f = torch.rand(1, 96, 20, 28, generator=g) * 2 - 1
        rec["feat_in"] = f.numpy()
        rec["rec_ctx_ds_out"] = (model.rec_ctx_ds(f) + f).numpy()
        rec["conv_flow_bottom_out"] = model.vfinet.conv_flow_bottom(f).numpy()
        rec["conv_flow1_out"] = model.vfinet.conv_flow1(f).numpy()
        f100 = torch.rand(1, 100, 20, 28, generator=g) * 2 - 1
        rec["flow2_in"] = f100.numpy()
        rec["conv_flow2_out"] = model.vfinet.conv_flow2(f100).numpy()
    np.savez_compressed(out_path, **rec)
    print("wrote", out_path, os.path.getsize(out_path) >> 10, "KiB")


def import_contract(out_path):
    """tests/golden/import_contract.json: for every module of the reference that fldr-vfi_amd/ shadows, the NAMES the
    reference's own files import from it (explicit `from m import a, b`) or use through `from m import *` — collected
    with `ast` from the import statements and free names of main.py, run_on_your_images.py, utils.py, fLDRnet.py,
    useful.py and OpticalFlow/PWCNet.py.  Names only; no reference code is copied."""
    import ast
    import builtins
    import json
    shadowed = ("pca_comp", "useful", "fLDRnet", "softSplat", "OpticalFlow", "OpticalFlow.PWCNet", "OpticalFlow.correlation")
    files = ("main.py", "run_on_your_images.py", "utils.py", "fLDRnet.py", "useful.py", "OpticalFlow/PWCNet.py")

    def public_toplevel(path):
        names = set()
        for n in ast.parse(open(path).read()).body:
            if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                names.add(n.name)
            elif isinstance(n, ast.Import):
                names.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, ast.ImportFrom):
                names.update(a.asname or a.name for a in n.names)
            elif isinstance(n, (ast.Assign, ast.AnnAssign, ast.AugAssign)):
                names.update(t.id for t in ast.walk(n) if isinstance(t, ast.Name) and isinstance(t.ctx, ast.Store))
        return {x for x in names if not x.startswith("_")}

    contract = {m: {"names": set(), "star_names": set(), "importers": set()} for m in shadowed}
    for f in files:
        tree = ast.parse(open(os.path.join(REF, f)).read())
        pkg = os.path.dirname(f).replace("/", ".")
        bound, stars = set(), []
        for n in ast.walk(tree):
            if isinstance(n, ast.ImportFrom):
                mod = n.module or ""
                if n.level:                                           # `from . import correlation` inside OpticalFlow/
                    mod = pkg + ("." + mod if mod else "")
                for a in n.names:
                    if a.name == "*":
                        stars.append(mod)
                    else:
                        bound.add(a.asname or a.name)
                        if mod in contract:
                            contract[mod]["names"].add(a.name)
                            contract[mod]["importers"].add(f)
            elif isinstance(n, ast.Import):
                bound.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                bound.add(n.name)
            elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                bound.add(n.id)
            elif isinstance(n, ast.arg):
                bound.add(n.arg)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                bound.add(n.name)
        free = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)} - bound - set(dir(builtins))
        for mod in stars:
            if mod in contract:
                used = free & public_toplevel(os.path.join(REF, mod.replace(".", "/") + ".py"))
                contract[mod]["star_names"].update(used)
                contract[mod]["importers"].add(f)
    out = {m: {k: sorted(v) for k, v in c.items()} for m, c in contract.items() if c["names"] or c["star_names"]}
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", out_path, {m: len(c["names"]) + len(c["star_names"]) for m, c in out.items()})


# --test3scales / --test4scales / --test6scales / --test7scales (main.py:243-268): what main.py does to the namespace after
# --papermodel, then one small forward of the reference at that pyramid depth (frames + output frame only).
DEPTH_CASES = {3: (100, 150, 0.5, 21), 4: (128, 200, 0.25, 22), 6: (300, 400, 0.5, 23), 7: (520, 530, 0.75, 24)}   # reflect padding needs H, W > half the unit
DEPTH_LISTS = {3: ([8, 16, 32, 64], [4, 16, 64, 256]),
               4: ([8, 16, 32, 64, 128], [4, 16, 64, 256, 1024]),
               6: ([8, 16, 32, 64, 128, 256, 512], [4, 16, 64, 256, 1024, 4096, 16384]),
               7: ([8, 16, 32, 64, 128, 256, 512, 1024], [4, 16, 64, 256, 1024, 4096, 16384, 65536])}


def depth_cases(R, fLDRnet, pca_comp, args, gold):
    import copy
    for S, (H, W, tval, seed) in DEPTH_CASES.items():
        a = copy.copy(args)
        a.scales, a.fractions = (list(v) for v in DEPTH_LISTS[S])
        a.S_tst = S
        a.moreTstSc = S != 3
        a.phase = "test"
        model, _ = load_model(fLDRnet, pca_comp, a)
        # one DCTParams per level (run_on_your_images.py:66 builds six; save_params only takes the first list it is given)
        model.params = [pca_comp.DCTParams(8, 0.25, 0.5) for _ in range(S + 1)]
        u8 = O.synthetic_pair(H, W, seed=seed, quadrant=True)
        frames = O.frames_from_uint8(u8)
        B, C, T, _, _ = frames.shape
        div = (2 ** a.S_tst) * 8                                                   # main.py:842
        ph, pw = (div - H % div) % div, (div - W % div) % div
        x = torch.nn.functional.pad(frames.reshape(B, -1, H, W), (0, pw, 0, ph), a.padding).reshape(B, C, T, H + ph, W + pw)
        Bp, Cp, Tp, Hp, Wp = x.shape
        pyr = [torch.nn.functional.interpolate(                                       # main.py:855-856
            x.permute(0, 2, 1, 3, 4).reshape(Bp * Tp, Cp, Hp, Wp), scale_factor=a.scales[0] / a.scales[i],
            mode="bicubic", align_corners=a.align_cornerse).reshape(
                Bp, Tp, Cp, int(Hp * (a.scales[0] / a.scales[i])), int(Wp * (a.scales[0] / a.scales[i]))
            ).permute(0, 2, 1, 3, 4) if i != 0 else x for i in range(a.S_tst + 1)]
        t = torch.tensor([[tval]], dtype=torch.float32)
        with torch.no_grad():
            out, _ = model([torch.zeros(B, 96, Hp // 8, Wp // 8) for _ in range(S + 1)], t, normInput=[p.clone() for p in pyr], is_training=False, validation=False)
        out = out[:, :, :H, :W]
        path = os.path.join(gold, "depth_S%d_%dx%d.npz" % (S, H, W))
        # the whole frame for the small cases; for the deep pyramids (large padded frames) a 160 x 240 window from the middle
        # plus per-channel sums of the whole frame
        y0, x0 = (0, 0) if S < 6 else ((H - 160) // 2, (W - 240) // 2)
        win = out if S < 6 else out[..., y0:y0 + 160, x0:x0 + 240]
        np.savez_compressed(path, frames_u8=u8.numpy(), t=np.float32(tval), S_tst=np.int32(S), padded=np.int32([Hp, Wp]),
                            out=win.float().numpy(), window=np.int32([y0, x0, win.shape[-2], win.shape[-1]]),
                            out_sum=out.sum((0, 2, 3)).numpy(), out_abssum=out.abs().sum((0, 2, 3)).numpy())
        print("wrote", path, os.path.getsize(path) >> 10, "KiB  padded", (Hp, Wp))


def main():
    if "--contract-only" in sys.argv:
        import_contract(os.path.join(ROOT, "tests", "golden", "import_contract.json"))
        return
    if "--depths-only" in sys.argv:
        if sys.flags.optimize < 1:
            raise SystemExit("run with python -O (fLDRnet.py:448 asserts a CUDA device index)")
        torch.manual_seed(0)
        torch.set_num_threads(8)
        R, fLDRnet, pca_comp, args = import_reference()
        depth_cases(R, fLDRnet, pca_comp, args, os.path.join(ROOT, "tests", "golden"))
        return
    if sys.flags.optimize < 1:
        raise SystemExit("run with python -O (fLDRnet.py:448 asserts a CUDA device index)")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R, fLDRnet, pca_comp, args = import_reference()
    model, ckpt = load_model(fLDRnet, pca_comp, args)
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    wdir = os.path.join(ROOT, "fldr-vfi_amd", "weights")
    os.makedirs(wdir, exist_ok=True)
    export_weights(ckpt, os.path.join(wdir, "fLDRnet_X4K1000FPS_exp1_best_PSNR.npz"))
    rec = model_case(model, R, fLDRnet, args, 256, 256, 0.5, 0, False, os.path.join(gold, "model_256x256_t0500.npz"))
    model_case(model, R, fLDRnet, args, 200, 500, 0.125, 1, True, os.path.join(gold, "model_200x500_t0125.npz"), slim=True)
    identity_splat_case(model, args, rec, os.path.join(gold, "flows_identity_splat_256x256.npz"))
    op_cases(model, fLDRnet, pca_comp, args, os.path.join(gold, "ops.npz"))
    # args namespace the model reads -> fixture for the host-side config test
    names = ("img_ch dctvfi_nf nf scales fractions S_tst S_trn phase ref_feat_extrac optimizeEV allImUp "
             "noEVOptimization meanVecParam ExacOneEV simpleEVs sminterp sminterpInpIm noResidAddup impmasksoftsplat "
             "cutoffUnnec tempbottomflowfix align_cornerse outMaskLess TOptimization testgetflowout timetest "
             "mean_vector_norm padding patch_size validation_patch_size oneEV pcanet").split()
    import json
    with open(os.path.join(gold, "state_dict_keys.json"), "w") as f:
        json.dump({k: [list(v.shape), str(v.dtype)] for k, v in ckpt["state_dict_Model"].items()}, f, indent=0)
    from OpticalFlow.PWCNet import PWCNet as RefPWC          # reference class, for its state-dict layout only
    with open(os.path.join(gold, "pwcnet_state_dict_keys.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in RefPWC().state_dict().items()}, f, indent=0)
    with open(os.path.join(gold, "args_papermodel_test5scales.json"), "w") as f:
        json.dump({n: getattr(args, n) for n in names if hasattr(args, n)}, f, indent=1, sort_keys=True)
    import_contract(os.path.join(gold, "import_contract.json"))
    depth_cases(R, fLDRnet, pca_comp, args, gold)


if __name__ == "__main__":
    main()
