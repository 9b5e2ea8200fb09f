"""CPU-side checks (no GPU): the C ABI library loads and exports everything include/fldr_hip.h declares, the
host mirror of the reference interface (args, state-dict keys, padding/pyramid, error behaviour) is right."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import fldr_hip
    hdr = open(os.path.join(ROOT, "include", "fldr_hip.h")).read()
    declared = set(re.findall(r"\b(fldr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fldr_conv_desc"}
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(fldr_hip.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libfldr_hip.so does not export " + name
    assert declared == set(fldr_hip.EXPORTS), (declared ^ set(fldr_hip.EXPORTS))
    # the product library exports the integration ABI and nothing else: no tuning / cross-check hook
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", fldr_hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    # (-fvisibility=hidden + csrc/exports.map: the dynamic symbol table is the header, no mangled internals, no kernel handles)
    exported = set(l.split()[-1] for l in syms.splitlines() if l.strip())
    assert not [n for n in exported if n.startswith("fldr_debug_")], sorted(exported)
    assert not [n for n in exported if not n.startswith("fldr_")], sorted(n for n in exported if not n.startswith("fldr_"))[:10]
    assert exported == declared, sorted(exported ^ declared)
    # ... the TEST build adds exactly the hooks of include/fldr_hip_test_hooks.h (minus those of the stamp builds)
    hooks_hdr = open(os.path.join(ROOT, "include", "fldr_hip_test_hooks.h")).read()
    hooks_decl = set(re.findall(r"\b(fldr_[a-z0-9_]+)\s*\(", hooks_hdr))        # fldr_debug_* and the retired fldr_softsplat_tile*
    assert not [n for n in exported if n.startswith("fldr_softsplat_tile") and n != "fldr_softsplat_tile_ws_floats"], sorted(exported)
    tsyms = subprocess.run(["nm", "-D", "--defined-only", fldr_hip.TEST_LIB_PATH], capture_output=True, text=True, check=True).stdout
    texported = set(l.split()[-1] for l in tsyms.splitlines() if l.strip())
    assert not [n for n in texported if not n.startswith("fldr_")], sorted(n for n in texported if not n.startswith("fldr_"))[:10]
    assert declared <= texported and texported - declared <= hooks_decl, sorted(texported - declared - hooks_decl)
    assert set(fldr_hip.HOOKS) <= texported, sorted(set(fldr_hip.HOOKS) - texported)
    assert fldr_hip.lib().fldr_version() == 105 == fldr_hip.ABI_VERSION
    assert re.search(r"#define FLDR_VERSION 105\b", hdr)
    assert fldr_hip.lib().fldr_error_string(-2) == b"fldr: shape constraint violated"
    assert b"status block" in fldr_hip.lib().fldr_error_string(-3)


def test_conv_desc_layout_matches_header():
    import fldr_hip
    # 12 ptr + 12 i64 + 12 i32 + 12 i32 + i32 (+pad) + 4 ptr + 12 i32 + 1 ptr + 12 i64
    assert ctypes.sizeof(fldr_hip.ConvDesc) == 12 * 8 + 12 * 8 + 12 * 4 + 12 * 4 + 8 + 4 * 8 + 12 * 4 + 8 + 12 * 8
    # the ctypes mirrors against the structs the library was compiled with
    assert ctypes.sizeof(fldr_hip.ConvDesc) == fldr_hip.lib().fldr_sizeof_desc(0)
    assert ctypes.sizeof(fldr_hip.SpkConvDesc) == fldr_hip.lib().fldr_sizeof_desc(1)
    assert ctypes.sizeof(fldr_hip.PrepDesc) == fldr_hip.lib().fldr_sizeof_desc(2)
    assert ctypes.sizeof(fldr_hip.PcaLevel) == fldr_hip.lib().fldr_sizeof_desc(3) == 48       # raw_ws grew it from 40 (ABI 102)
    assert ctypes.sizeof(fldr_hip.SplatAccDesc) == fldr_hip.lib().fldr_sizeof_desc(4)
    assert ctypes.sizeof(fldr_hip.SplatGatherDesc) == fldr_hip.lib().fldr_sizeof_desc(5)
    assert fldr_hip.lib().fldr_sizeof_desc(6) < 0
    assert fldr_hip.lib().fldr_conv_prepack_size(96, 100, 3) == 104 * 9 * 96
    assert fldr_hip.lib().fldr_conv_prepack_size(6, 16, 3) == 16 * 9 * 16
    assert fldr_hip.lib().fldr_conv_prepack_size(16, 26, 4) == 28 * 272      # channel rows padded to 16 mod 32 floats
    assert fldr_hip.lib().fldr_conv_prepack_size(128, 16, 3) < 0


def test_args_match_reference_namespace():
    import fldr_harness as Hn
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "args_papermodel_test5scales.json")))
    a = Hn.args_config()
    for k, v in ref.items():
        assert getattr(a, k) == v, (k, getattr(a, k), v)


def test_state_dict_keys_and_strict_load():
    import fldr_harness as Hn
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
    a = Hn.args_config()
    model = a.net_object(a)
    own = model.state_dict()
    assert set(own) == set(ref), set(own) ^ set(ref)
    for k, (shape, dtype) in ref.items():
        assert list(own[k].shape) == shape and str(own[k].dtype) == dtype, k
    # the re-exported archive loads strictly once the aliases / unused entries are restored
    sd = Hn.npz_state_dict()
    missing = set(ref) - set(sd)
    assert all(re.match(r"(EV|Mean|meanVec)(16|32|64)$", k) or ".refine_unet.conv" in k for k in missing), missing
    model2, _, _ = Hn.prepare_model(torch.device("cpu"))
    assert float(model2.vfinet.T_param[0]) == pytest.approx(1.5616, abs=1e-3)
    assert model2.EV8.dtype == torch.float64 and model2.rec_ctx_ds[0].weight.dtype == torch.float32
    assert model2.base_modules[0] is model2.rec_ctx_ds and model2.base_modules[1] is model2.vfinet


def test_pad_and_pyramid_match_oracle_and_golden(oracle, golden):
    import fldr_harness as Hn
    g = golden("model_200x500_t0125")
    fr = Hn.frames_from_uint8(torch.from_numpy(g["frames_u8"]))
    a = Hn.args_config()
    pyr = Hn.build_pyramid(Hn.pad_frames(fr, a), a)
    assert pyr[0].shape == (1, 3, 2, 256, 512)
    ref = oracle.pad_and_pyramid(fr)
    for i in range(6):
        assert torch.equal(pyr[i], ref[i])
        if i:
            assert np.array_equal(pyr[i].numpy(), g["pyr%d" % i])
    u = Hn.synthetic_pair(200, 500, seed=1, quadrant=True)
    assert torch.equal(u, torch.from_numpy(g["frames_u8"])) and torch.equal(u, oracle.synthetic_pair(200, 500, 1, True))


@pytest.mark.parametrize("S", [3, 4, 6, 7])
def test_pyramid_depth_configs(oracle, golden, S):
    """args_config(test_scales=S) = what main.py builds under --papermodel --test<S>scales (main.py:240-273); the harness pads
    to 2^S * 8 and builds S + 1 levels exactly as the oracle (and the reference fixture) does."""
    import fldr_harness as Hn
    name = {3: "depth_S3_100x150", 4: "depth_S4_128x200", 6: "depth_S6_300x400", 7: "depth_S7_520x530"}[S]
    g = golden(name)
    a = Hn.args_config(test_scales=S)
    assert a.S_tst == S and len(a.scales) == S + 1 == len(a.fractions) and a.scales[0] == 8 and a.dctvfi_nf == 16
    assert a.scales == [8 * 2 ** i for i in range(S + 1)] and a.fractions == [4 ** (i + 1) for i in range(S + 1)]
    assert a.moreTstSc == (S != 3) and a.phase == "test"
    fr = Hn.frames_from_uint8(torch.from_numpy(g["frames_u8"]))
    pyr = Hn.build_pyramid(Hn.pad_frames(fr, a), a)
    ref = oracle.pad_and_pyramid(fr, n_levels=S + 1)
    assert len(pyr) == S + 1 and list(pyr[0].shape[-2:]) == list(g["padded"])
    for i in range(S + 1):
        assert torch.equal(pyr[i], ref[i])
    with pytest.raises(ValueError):
        Hn.args_config(test_scales=2)
    m = a.net_object(a)                                    # the model builds at any depth (the parameters do not depend on it)
    assert set(m.state_dict()) == set(Hn.args_config().net_object(Hn.args_config()).state_dict())


def test_cpu_tensors_fail_loudly():
    """No CPU fallback anywhere: same behaviour as the reference (softSplat.py:251-252, correlation.py:343-344)."""
    import softSplat
    from OpticalFlow import correlation
    import pca_comp
    import fldr_harness as Hn
    x = torch.zeros(1, 3, 8, 8)
    f = torch.zeros(1, 2, 8, 8)
    with pytest.raises(NotImplementedError):
        softSplat.Softsplat()(x, f)
    with pytest.raises(NotImplementedError):
        softSplat._FunctionSoftsplat.apply(x, f)
    with pytest.raises(NotImplementedError):
        correlation.FunctionCorrelation(x, x)
    with pytest.raises(AssertionError):
        softSplat.FunctionSoftsplat(x, f, torch.zeros(1, 2, 8, 8), 'softmax')       # softSplat.py:321
    with pytest.raises(AssertionError):
        softSplat.FunctionSoftsplat(x, f, None, 'bogus')                             # softSplat.py:322
    model, _, a = Hn.prepare_model(torch.device("cpu"))
    with pytest.raises(NotImplementedError):
        pca_comp.to_pca_diff(torch.zeros(6, 16, 16), model.params[0], a, model.Mean8, model.EV8, model.meanVec8)
    with pytest.raises(NotImplementedError):
        model([None] * 6, torch.tensor([[0.5]]), normInput=[], is_training=True)


def test_missing_library_is_an_error(monkeypatch):
    import fldr_hip
    monkeypatch.setattr(fldr_hip, "_lib", None)
    monkeypatch.setattr(fldr_hip, "LIB_PATH", "/nonexistent/libfldr_hip.so")
    with pytest.raises(ImportError):
        fldr_hip.lib()


def test_psnr_and_uint8(oracle):
    import fldr_harness as Hn
    p = torch.tensor([[[-1.0, 1.0], [0.0, 3.0]]]).repeat(3, 1, 1)
    img = Hn.to_uint8_image(p)
    assert img.shape == (2, 2, 3) and img[0, 0, 0] == 0 and img[0, 1, 0] == 255 and img[1, 0, 0] == 128 and img[1, 1, 0] == 255
    assert Hn.psnr(img, img) == float("inf")
    assert Hn.psnr(img, img + 5) == pytest.approx(oracle.psnr(img, img + 5))


def test_pwcnet_state_dict_layout_matches_reference():
    from OpticalFlow.PWCNet import PWCNet
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "pwcnet_state_dict_keys.json")))
    own = PWCNet().state_dict()
    assert set(own) == set(ref), set(own) ^ set(ref)
    for k, shape in ref.items():
        assert list(own[k].shape) == shape, k


def test_import_contract_of_the_reference_drivers():
    """Every name the reference's own files import from a module this package shadows (collected with `ast` by
    tools/make_golden.py from main.py:12-18, run_on_your_images.py:9-15, utils.py:20-26, fLDRnet.py:16-22, useful.py:104,
    OpticalFlow/PWCNet.py:4) resolves here; `import *` names must survive an `__all__`."""
    import importlib
    contract = json.load(open(os.path.join(ROOT, "tests", "golden", "import_contract.json")))
    assert {"pca_comp", "useful", "fLDRnet", "softSplat"} <= set(contract)
    for mod, c in contract.items():
        m = importlib.import_module(mod)
        assert os.path.dirname(os.path.abspath(m.__file__)).startswith(os.path.join(ROOT, "fldr-vfi_amd")), (mod, m.__file__)
        for name in c["names"]:
            assert hasattr(m, name), "%s does not define %s (imported by %s)" % (mod, name, c["importers"])
        ns = {}
        exec("from %s import *" % mod, ns)
        for name in c["star_names"]:
            assert name in ns, "`from %s import *` does not provide %s" % (mod, name)
    # the exact statements of the drivers, rebuilt from the fixture
    ns = {}
    for mod in ("pca_comp", "useful", "softSplat"):
        exec("from %s import %s" % (mod, ",".join(contract[mod]["names"])), ns)
    assert ns["DCTParams"](8, 0.25, 0.5).wiS == 8


def test_training_only_names_raise_when_called_not_at_import():
    import pca_comp
    import useful
    with pytest.raises(NotImplementedError):
        pca_comp.to_pca(np.zeros((3, 16, 16)), pca_comp.DCTParams(8, 0.25, 0.5))
    with pytest.raises(NotImplementedError):
        pca_comp.pca_inverse(torch.zeros(1, 48, 2, 2), None, [], 16)
    # the small training-time helpers of useful.py are real (plain torch, off the hot path)
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    s = useful.ScaleIt("x", x, 2)
    y = s.scale(x)
    assert y.dtype == torch.float32 and float(y.amin()) == 0.0 and float(y.amax()) == 1.0
    assert torch.allclose(s.backscale(y), x, atol=1e-4)
    g = torch.Generator().manual_seed(0)
    d = torch.randn(200, 8, generator=g, dtype=torch.float64) @ torch.randn(8, 8, generator=g, dtype=torch.float64)
    p = useful.MYPCA(n_components=3)
    r = p.fit_transform(d.clone(), "cpu")
    assert r.shape == (200, 3) and torch.allclose(p.eigenvectors @ p.eigenvectors.T, torch.eye(3, dtype=torch.float64), atol=1e-10)
    full = useful.MYPCA()
    assert torch.allclose(full.inverse_transform(full.fit_transform(d.clone(), "cpu")), d, atol=1e-9)
    fl = [torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 8, 8)]
    assert float(useful.distillation_loss(fl, torch.zeros(1, 4, 32, 32), "cpu")) >= 0.0


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_reference_driver_imports_against_this_package():
    """The drop-in claim of INTEGRATION.md, executed: with fldr-vfi_amd/ first on the path, the reference's
    run_on_your_images.py (its import block and args_config) and main.py's import block run against THIS package's
    fLDRnet / softSplat / pca_comp / useful / OpticalFlow (third-party packages this image lacks are inert placeholders)."""
    import subprocess
    import sys
    code = r'''
import sys, types, ast
class Inert(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"): raise AttributeError(n)
        m = Inert(self.__name__ + "." + n); setattr(self, n, m); return m
    def __call__(self, *a, **k): return Inert("call")
for n in ("cupy", "cv2", "skimage", "skimage.feature", "skimage.metrics", "skimage.transform", "torchvision",
          "torchvision.transforms", "torchvision.models", "torchvision.utils", "torch.utils.tensorboard"):
    sys.modules[n] = Inert(n)
sys.path[:0] = [%r, "/root/reference"]
sys.argv = ["x"]
import run_on_your_images as R
import fLDRnet, softSplat, pca_comp, useful
for m in (fLDRnet, softSplat, pca_comp, useful):
    assert "fldr-vfi_amd" in m.__file__, m.__file__
assert R.DCTXVFInet is fLDRnet.DCTXVFInet and R.to_pca is pca_comp.to_pca
a = R.args_config()
assert a.net_object is fLDRnet.DCTXVFInet and a.S_tst == 5 and a.dctvfi_nf == 16
# main.py: execute only its import block (the module body parses the command line and starts a run)
src = open("/root/reference/main.py").read()
tree = ast.parse(src)
imports = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
ns = {}
exec(compile(ast.Module(imports, []), "main_imports", "exec"), ns)
assert ns["DCTXVFInet"] is fLDRnet.DCTXVFInet and ns["ScaleIt"] is useful.ScaleIt and ns["Softsplat"] is softSplat.Softsplat
model = a.net_object(a)
print("OK", type(model).__module__)
''' % os.path.join(ROOT, "fldr-vfi_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK fLDRnet" in r.stdout, r.stdout + r.stderr


def test_markstein_quotient_is_the_correctly_rounded_fp64_division():
    """pca_pyramid_kernels.hip divides by wave-uniform values with q = x*r, e = fma(-q, c, x), q' = fma(e, r, q) on
    r = RN(1/c) instead of the IEEE division sequence.  Exact rational arithmetic (float(Fraction) rounds to nearest
    even) shows q' == RN(x / c) for the divisors this path meets: the 16 meanVec8 entries of the shipped checkpoint and
    min/max ranges of projected features, over random and adversarial numerators."""
    from fractions import Fraction as Fr
    import random
    z = np.load(os.path.join(ROOT, "fldr-vfi_amd", "weights", "fLDRnet_X4K1000FPS_exp1_best_PSNR.npz"))
    divisors = [float(v) for v in z["meanVec8"]] + [3.0, 7.345678901234567, 1.0 / 3.0, 41.70302134, 0.1, 1e-3 + 1e-19, 123456.789]

    def fma(a, b, c):
        return float(Fr(a) * Fr(b) + Fr(c))

    rng = random.Random(5)
    checked = 0
    for c in divisors:
        r = float(Fr(1) / Fr(c))                      # RN(1 / c)
        xs = [rng.uniform(-40.0, 40.0) for _ in range(400)] + [rng.uniform(-1e-3, 1e-3) for _ in range(100)]
        xs += [c * k for k in (1.0, 3.0, 0.5, 1 + 2 ** -52, 1 - 2 ** -53)] + [float(np.nextafter(c * 5.0, 0.0)), float(np.nextafter(c * 5.0, 100.0))]
        for x in xs:
            q = float(Fr(x) * Fr(r))
            e = fma(-q, c, x)
            q2 = fma(e, r, q)
            assert q2 == float(Fr(x) / Fr(c)), (x, c, q2)
            checked += 1
    assert checked > 10000


def test_synthetic_pairs_are_seeded_and_differ_by_motion_model():
    """bench.py's workloads: the global-shift pairs of the headline and the varying-motion pairs of its `varying_motion`
    record are deterministic functions of the seed, uint8 [2,3,H,W], and the second frame really moves."""
    import torch
    import fldr_harness as Hn
    a, b = Hn.synthetic_pair(64, 96, seed=3), Hn.synthetic_pair(64, 96, seed=3)
    v, w = Hn.synthetic_pair_varying(64, 96, seed=3), Hn.synthetic_pair_varying(64, 96, seed=3)
    for x in (a, v):
        assert x.shape == (2, 3, 64, 96) and x.dtype == torch.uint8
    assert torch.equal(a, b) and torch.equal(v, w)
    assert not torch.equal(Hn.synthetic_pair(64, 96, seed=4), a)
    assert (v[0].float() - v[1].float()).abs().mean().item() > 1.0 and (a[0].float() - a[1].float()).abs().mean().item() > 1.0


def _write_scene(folder, n_frames, h=24, w=40, seed=0):
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    g = np.random.default_rng(seed)
    base = g.integers(0, 256, size=(h, w + n_frames, 3), dtype=np.uint8)
    paths = []
    for k in range(n_frames):
        bgr = base[:, k:k + w]                                             # a scene sliding by one pixel per frame
        Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(os.path.join(folder, "%05d.png" % k))   # PNG stores RGB
        paths.append(bgr)
    return paths


def test_evaluate_dir_walk_sharding_and_reduction(tmp_path):
    """fldr_harness.evaluate_dir on a 3-triplet X-Test-style folder written here (main.py:815-911, utils.py:414-432): the sample list
    is the reference's (pairs t_step_size apart, targets in between, grouped by pair), frames come back in cv2's BGR order, the
    PSNR mean is the mean over triplets, and the shares of two ranks add up to the single-rank result.  The model is replaced by
    a stub (mean of the two frames) so that this runs without a GPU; the GPU suite runs the real thing."""
    import fldr_harness as Hn
    root = str(tmp_path)
    a = _write_scene(os.path.join(root, "Type1", "TEST01"), 5, seed=1)     # pairs (0,2), (2,4): 2 triplets at multiple=2
    b = _write_scene(os.path.join(root, "Type2", "TEST02"), 3, seed=2)     # pair (0,2): 1 triplet
    pairs = Hn.list_xtest_triplets(root, multiple=2, t_step_size=2)
    assert [(os.path.basename(p0), os.path.basename(p1), sc, [(os.path.basename(t), tv) for t, tv in tg]) for p0, p1, sc, tg in pairs] == [
        ("00000.png", "00002.png", "Type1/TEST01", [("00001.png", 0.5)]), ("00002.png", "00004.png", "Type1/TEST01", [("00003.png", 0.5)]),
        ("00000.png", "00002.png", "Type2/TEST02", [("00001.png", 0.5)])]
    assert np.array_equal(Hn.load_bgr_u8(pairs[0][0]), a[0])              # BGR, as cv2.imread
    p8 = Hn.list_xtest_triplets(os.path.join(root, "none"), 8, 32)
    assert p8 == []
    with pytest.raises(RuntimeError):
        Hn.evaluate_dir(os.path.join(root, "none"), predict=lambda *x: [])

    def stub(frames_u8, ts, targets):
        pred = frames_u8.float().mean(1).round()                           # [1,3,H,W]
        return [(Hn.psnr(t[0].permute(1, 2, 0).numpy(), pred[0].permute(1, 2, 0).numpy()), 0.5) for t in targets]
    full = Hn.evaluate_dir(root, multiple=2, t_step_size=2, predict=stub)
    want = [Hn.psnr(tgt, np.round((x0.astype(np.float32) + x1.astype(np.float32)) / 2)) for x0, x1, tgt in ((a[0], a[2], a[1]), (a[2], a[4], a[3]), (b[0], b[2], b[1]))]
    assert full["frames"] == 3 and full["pairs"] == 3 and full["psnr"] == pytest.approx(sum(want) / 3, rel=1e-12) and full["ssim"] == 0.5
    assert full["per_t"] == {0.5: pytest.approx(sum(want) / 3)}
    r0 = Hn.evaluate_dir(root, multiple=2, t_step_size=2, predict=stub, rank=0, world=2)    # no process group: each rank's own share
    r1 = Hn.evaluate_dir(root, multiple=2, t_step_size=2, predict=stub, rank=1, world=2)
    assert r0["frames"] == 2 and r1["frames"] == 1
    assert (r0["psnr"] * 2 + r1["psnr"] * 1) / 3 == pytest.approx(full["psnr"], rel=1e-12)
    assert Hn.reduce_sums([1.5, 2.0, 3.0]) == [1.5, 2.0, 3.0]


def test_dec23_softmax_exp_polynomial_is_accurate_to_1e12():
    """dec23_kernels.hip::d23_exp_nonpos (the fused synthesis kernel's exp for softmax arguments <= 0): its constants, read from the source,
    evaluated in numpy fp64 the way the kernel evaluates them — relative error below 1e-12 over [-745, 0] (the fp32 logits carry ~6e-8),
    exact 1 at 0, 0 beyond the denormals."""
    import math
    src = open(os.path.join(ROOT, "fldr-vfi_amd", "csrc", "dec23_kernels.hip")).read()
    body = src[src.index("double d23_exp_nonpos(double x) {"):]
    body = body[:body.index("return __builtin_ldexp")]
    body = re.sub(r"//[^\n]*", "", body)                                 # (the comments name the factorials)
    nums = [float(v) for v in re.findall(r"-?\d+\.\d+(?:e[-+]\d+)?", body)]
    assert abs(nums[0] - 1.4426950408889634) < 1e-15 and abs(nums[1] + math.log(2)) < 1e-9 and abs(nums[2]) < 1e-9
    coef = nums[3:]
    assert len(coef) == 11 and coef[-1] == coef[-2] == 1.0 and coef[-3] == 0.5
    for k, c in zip(range(10, 0, -1), coef[:-1]):
        assert abs(c * math.factorial(k) - 1.0) < 1e-15                  # Taylor coefficients 1 / k!
    x = np.concatenate([-np.abs(np.random.default_rng(0).standard_normal(400000)) * 30, -np.linspace(0, 745, 100001)])
    n = np.rint(x * nums[0])
    r = x + n * nums[1]
    r = r + n * nums[2]
    pl = np.full_like(x, coef[0])
    for c in coef[1:]:
        pl = pl * r + c
    got, ref = np.ldexp(pl, n.astype(np.int64)), np.exp(x)
    m = ref > 1e-300
    assert np.max(np.abs(got[m] - ref[m]) / ref[m]) < 1e-12
    assert got[np.argmax(x)] == 1.0 and np.all(np.ldexp(pl[:4], np.full(4, -1200)) == 0.0)


def test_dec23_counted_vmcnt_invariant(tmp_path):
    """fldr_dec23_synth's consumer waves prove that their LDS-DMA pieces of tile k + 2 have landed with `s_waitcnt vmcnt(3)` in front of
    the tile barrier — correct only if at least three vector-memory instructions (the frame stores) follow the last piece on EVERY path,
    free only if exactly three do and the compiler adds no vmcnt(0) of its own in between (round 5's build did: the stores sat in a
    divergent region and the candidates' loads were still pending behind the pieces).  Nothing in the source enforces the emitted code,
    so the gfx950 listing is checked (tools/check_dma_waits.py: dataflow over the kernel's basic blocks): the two product instantiations
    (fp64 frame, 8-bit frame) reach the counted wait with exactly three stores behind the pieces on every path, no barrier is reached with
    an uncovered piece, no scratch memory; the fp32-output instantiation (tests only) keeps a compiler-placed vmcnt(0) — safe, not counted."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "fldr-vfi_amd", "csrc", "dec23_kernels.hip")
    out = str(tmp_path / "dec23.s")
    subprocess.run([hipcc, "@" + os.path.join(ROOT, "fldr-vfi_amd", "csrc", "hipcc_flags.rsp"), "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True, timeout=600)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dma_waits as C
    for inst in ("dec23_synth_kernelIdE", "dec23_synth_kernelIhE"):
        problems, notes = C.check(out, inst, expect_counted=3)
        assert not problems, (problems, notes)
        assert "[(3, [3])]" in notes[0], notes
    problems, notes = C.check(out, "dec23_synth_kernelIfE")
    assert not problems, (problems, notes)


def test_product_kernels_use_no_scratch():
    """Every gfx950 kernel of the PRODUCT library runs without scratch memory and without spilled vector registers (AMDGPU metadata of the
    code objects embedded in libfldr_hip.so, tools/kernel_resources.py).  A kernel that starts spilling fails no numerical test — it gets
    slower, and a spill reload is an `s_waitcnt vmcnt(0)` in the middle of a pipeline that counts its outstanding loads (the ring's and the
    synthesis kernel's LDS-DMA): round 6 found three ring kernels with 60-120 bytes of scratch (+18 %) only by running the previous
    round's build beside the new one on the same box."""
    import sys
    lib = os.path.join(ROOT, "fldr-vfi_amd", "libfldr_hip.so")
    if not os.path.exists(lib) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("library not built / no llvm-readelf")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as K
    ks = K.kernels(lib)
    assert len(ks) > 100, len(ks)
    assert any("conv3x3_ring_kernel" in k["name"] for k in ks) and any("dec23_synth_kernel" in k["name"] for k in ks)
    bad = [(k["name"], k.get("scratch"), k.get("vgpr_spills")) for k in ks if k.get("scratch", 0) or k.get("vgpr_spills", 0)]
    assert not bad, bad


def test_no_packed_fp32_source_read_through_op_sel_behind_another_vector_source():
    """gfx950: a packed-fp32 add / mul / fma whose first vector-register source is read straight and a LATER one through op_sel = 1 returns a wrong
    low half in lanes 48-63 now and then while waves of another kernel issue matrix instructions on the same SIMD (stand-alone reproducer:
    tools/ubench/pk_opsel_probe.hip, profiles/r06_pk_opsel_probe.txt).  hipcc forms such instructions on its own — one in level0_prep's tap-window
    build wrote runs of 16 wrong pixels whenever a convolution of another frame pair shared the CUs (profiles/r06_prep_concurrency.txt) — so the
    disassembly of BOTH built libraries is checked (tools/check_pk_opsel.py), and the libraries are compiled without packed fp32 instructions
    altogether (one flags file for every build recipe: csrc/hipcc_flags.rsp)."""
    import sys
    libs = [os.path.join(ROOT, "fldr-vfi_amd", n) for n in ("libfldr_hip.so", "libfldr_hip_test.so")]
    if not all(os.path.exists(l) for l in libs) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("libraries not built / no llvm-objdump")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_pk_opsel as C
    # the rule itself, on the forms the probe measured: affected ...
    lab = lambda l: None
    for line in ("v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]", "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]",
                 "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_fma_f32 v[0:1], v[2:3], 1.0, v[4:5] op_sel:[0,0,1] op_sel_hi:[1,0,1]",
                 "v_pk_fma_f32 v[0:1], s[2:3], v[4:5], v[6:7] op_sel:[0,0,1]"):
        assert C.offenders_in_text(["\t" + line], lab), line
    # ... and not affected (`op_sel:[1,1]` measured clean too; the rule flags it all the same: any later vector source through op_sel)
    for line in ("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]", "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]",
                 "v_pk_fma_f32 v[8:9], s[72:73], v[72:73], v[8:9] op_sel:[0,1,0]", "v_pk_add_f32 v[0:1], v[2:3], s[4:5] op_sel:[0,1]", "v_pk_mul_f32 v[0:1], s[2:3], v[4:5] op_sel:[1,0]",
                 "v_pk_mul_f32 v[2:3], s[20:21], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]", "v_pk_add_f32 v[0:1], v[2:3], v[4:5]"):
        assert not C.offenders_in_text(["\t" + line], lab), line
    for l in libs:
        assert C.offenders(l) == [], (l, C.offenders(l)[:5])
    # ... and since the end of round 6 NO packed fp32 arithmetic at all, in any kernel of either library (csrc/hipcc_flags.rsp: -target-feature
    # -packed-fp32-ops, passed by every build recipe; the forward is 1.4 % faster without it)
    import kernel_resources as K, subprocess, tempfile
    for lib in libs:
        n_kernels = 0
        for blob in K.code_objects(lib):
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob); f.flush()
                txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f.name], capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    n_kernels += 1
                else:
                    assert not re.search(r"\bv_pk_(add|mul|fma)_f32\b", line), (lib, cur, line)
        assert n_kernels > 100, (lib, n_kernels)
    assert "-packed-fp32-ops" in open(os.path.join(ROOT, "fldr-vfi_amd", "csrc", "hipcc_flags.rsp")).read()


def test_ring_kernels_listing_invariants(tmp_path):
    """What the same-box comparison with the round-5 kernels found in the convolution ring (profiles/r06_ring_wait_ab.txt), kept from
    coming back by a look at the gfx950 listing of the product build — nothing in the source enforces what the compiler emits:
      (a) no ring kernel uses scratch memory (a poison word read in every epilogue had put 14 spills into the residual kernels: +18 %);
      (b) the bounded wait is a SCALAR loop of eight polls per trip, each with its own exit: every `s_sleep` is followed by the poll's
          `ds_read_b32`, a `v_readfirstlane_b32`, a scalar compare and a scalar branch, and a kernel holds 16 of them (two wait sites)
          — the limit passed by reference through an early return had made the loop divergent (counter in a VGPR, exec-mask control);
      (c) the fault report (the atomic on fldr_ring_timeouts) is not inlined behind a wait: none within 80 lines after an `s_sleep`."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "fldr-vfi_amd", "csrc", "conv_ring_kernels.hip")
    out = str(tmp_path / "ring.s")
    subprocess.run([hipcc, "@" + os.path.join(ROOT, "fldr-vfi_amd", "csrc", "hipcc_flags.rsp"), "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True, timeout=600)
    L = open(out).read().splitlines()
    scratch, name = {}, None
    for l in L:
        m = re.search(r"\.amdhsa_kernel (\S+)", l)
        if m:
            name = m.group(1)
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", l)
        if m and name:
            scratch[name] = int(m.group(1))
    ring = [k for k in scratch if "conv3x3_ring" in k]
    assert len(ring) >= 40, len(ring)
    assert not [k for k in ring if scratch[k]], [(k, scratch[k]) for k in ring if scratch[k]]
    for k in ring:
        i0 = next(i for i, l in enumerate(L) if l.startswith(k + ":"))
        i1 = next(i for i in range(i0, len(L)) if L[i].startswith(".Lfunc_end"))
        body = [l for l in L[i0:i1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
        sleeps = [i for i, l in enumerate(body) if l.split()[0] == "s_sleep"]
        assert len(sleeps) == 16, (k, len(sleeps))
        for i in sleeps:
            ops = [l.split()[0] for l in body[i + 1:i + 9]]
            assert ops[0] == "ds_read_b32" and "v_readfirstlane_b32" in ops, (k, ops)
            j = ops.index("v_readfirstlane_b32")
            assert any(o.startswith("s_cmp_") for o in ops[j:]) and any(o.startswith("s_cbranch_scc") or o == "s_cselect_b64" for o in ops[j:]), (k, ops)
            assert not any(o.startswith("global_atomic") for o in (l.split()[0] for l in body[i:i + 80])), (k, "fault report inlined behind a wait")


def test_dma_wait_checker_detects_violations(tmp_path):
    """tools/check_dma_waits.py on hand-written listings: the analyser that guards the synthesis kernel's counted wait must itself flag
    (a) a counted wait with too few vector-memory instructions behind the last LDS-DMA piece on ONE of two paths, (b) a barrier reached
    with an uncovered piece, (c) scratch memory — and pass the sound form (three stores on every path, or a vmcnt(0))."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dma_waits as C

    def listing(name, body, scratch=0):
        return ("\t.amdhsa_kernel %s\n\t\t.amdhsa_private_segment_fixed_size %d\n\t.end_amdhsa_kernel\n%s:                ; @%s\n%s\ts_endpgm\n"
                % (name, scratch, name, name, body))
    good = listing("_Z4goodv", """\tglobal_load_lds_dwordx4 v[0:1], off
\tv_add_f32_e32 v2, v3, v4
\tglobal_store_dwordx4 v[0:1], v[4:7], off
\tglobal_store_dwordx4 v[0:1], v[4:7], off
\tglobal_store_dwordx4 v[0:1], v[4:7], off
\ts_waitcnt vmcnt(3) lgkmcnt(0)
\ts_barrier
""")
    two_paths = listing("_Z3badv", """\tglobal_load_lds_dwordx4 v[0:1], off
\ts_cbranch_execz .LBB0_2
\tglobal_store_dwordx4 v[0:1], v[4:7], off
\tglobal_store_dwordx4 v[0:1], v[4:7], off
\tglobal_store_dwordx4 v[0:1], v[4:7], off
.LBB0_2:
\ts_waitcnt vmcnt(3) lgkmcnt(0)
\ts_barrier
""")
    covered = listing("_Z7coveredv", """\tglobal_load_lds_dwordx4 v[0:1], off
\ts_cbranch_execz .LBB1_2
\tglobal_store_dwordx4 v[0:1], v[4:7], off
\tglobal_store_dwordx4 v[0:1], v[4:7], off
\tglobal_store_dwordx4 v[0:1], v[4:7], off
\ts_branch .LBB1_3
.LBB1_2:
\ts_waitcnt vmcnt(0)
.LBB1_3:
\ts_waitcnt vmcnt(3) lgkmcnt(0)
\ts_barrier
""")
    spilled = listing("_Z7spilledv", """\tglobal_load_lds_dwordx4 v[0:1], off
\ts_waitcnt vmcnt(0)
\ts_barrier
""", scratch=24)
    p = str(tmp_path / "k.s")
    open(p, "w").write(good + two_paths + covered + spilled)
    ok, notes = C.check(p, "_Z4goodv", expect_counted=3)
    assert not ok and "[(3, [3])]" in notes[0], (ok, notes)
    bad, _ = C.check(p, "_Z3badv", expect_counted=3)
    assert any("vmcnt(3) with only [0]" in x for x in bad) and any("s_barrier reached with an LDS-DMA piece pending" in x for x in bad), bad
    ok2, _ = C.check(p, "_Z7coveredv", expect_counted=3)
    assert not ok2, ok2
    sp, _ = C.check(p, "_Z7spilledv")
    assert any("scratch" in x for x in sp), sp
