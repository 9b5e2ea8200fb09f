"""Pin the CPU oracle against golden vectors captured from the reference's own
model code (tools/make_golden.py) and against known-answer tests for the two
CUDA-only operators."""
import math

import numpy as np
import pytest
import torch

CASES = ["model_256x256_t0500", "model_200x500_t0125"]


def _run(oracle, weights, g, keep):
    fr = oracle.frames_from_uint8(torch.from_numpy(g["frames_u8"]))
    pyr = oracle.pad_and_pyramid(fr)
    with torch.no_grad():
        out = oracle.forward(weights, pyr, torch.tensor([[float(g["t"])]]), keep=keep)
    return pyr, out


@pytest.mark.parametrize("case", CASES)
def test_full_forward_matches_reference(oracle, weights, golden, case):
    g = golden(case)
    keep = {}
    pyr, out = _run(oracle, weights, g, keep)
    assert out.dtype == torch.float64 and str(g["out_dtype"]) == "torch.float64"      # F3
    for i in range(1, 6):
        assert np.array_equal(pyr[i].numpy(), g["pyr%d" % i])
        np.testing.assert_allclose(keep["pca"][i].numpy(), g["pca%d" % i], atol=1e-6)
        np.testing.assert_allclose(keep["feat"][i].numpy(), g["feat%d" % i], atol=1e-5)
        np.testing.assert_allclose(keep["flows"][i].numpy(), g["flow%d" % i], atol=1e-4)
    cat = torch.cat([pyr[0][:, :, 0], pyr[0][:, :, 1], keep["warped0"], keep["warped1"], keep["flow_t0"],
                     keep["flow_t1"], keep["flowback_0"], keep["flowback_1"], keep["im0_tot"], keep["im1_tot"]], 1)
    np.testing.assert_allclose(cat.double().sum((0, 2, 3)).numpy(), g["cat26_sum"], rtol=1e-6, atol=1e-3)
    for ci, (y0, x0, h, w) in enumerate(g["crops"]):
        np.testing.assert_allclose(cat[..., y0:y0 + h, x0:x0 + w].numpy(), g["cat26_crops"][ci], atol=1e-5)
        np.testing.assert_allclose(keep["refine_out"][..., y0:y0 + h, x0:x0 + w].numpy(),
                                   g["refine_out_crops"][ci], atol=1e-4)
    if g["out"].dtype == np.float64:
        assert np.array_equal(out.numpy(), g["out"])            # the oracle reproduces the reference's fp64 frame bit for bit
    else:
        np.testing.assert_allclose(out.numpy(), g["out"], atol=1e-6)        # (fixture stored as fp32)


DEPTHS = ["depth_S3_100x150", "depth_S4_128x200", "depth_S6_300x400", "depth_S7_520x530"]


@pytest.mark.parametrize("case", DEPTHS)
def test_pyramid_depths_match_reference(oracle, weights, golden, case):
    """--test3scales / --test4scales / --test6scales / --test7scales (main.py:243-268): the oracle at S_tst + 1 levels against
    the reference's own forward at that depth (fixtures: tools/make_golden.py depth_cases)."""
    g = golden(case)
    S = int(g["S_tst"])
    fr = oracle.frames_from_uint8(torch.from_numpy(g["frames_u8"]))
    H, W = fr.shape[-2:]
    pyr = oracle.pad_and_pyramid(fr, n_levels=S + 1)
    assert list(pyr[0].shape[-2:]) == list(g["padded"]) and len(pyr) == S + 1
    with torch.no_grad():
        out = oracle.forward(weights, pyr, torch.tensor([[float(g["t"])]]))[..., :H, :W]
    y0, x0, h, w = (int(v) for v in g["window"])
    np.testing.assert_allclose(out[..., y0:y0 + h, x0:x0 + w].numpy(), g["out"], atol=1e-6)
    np.testing.assert_allclose(out.sum((0, 2, 3)).numpy(), g["out_sum"], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(out.abs().sum((0, 2, 3)).numpy(), g["out_abssum"], rtol=1e-9, atol=1e-6)


def test_identity_splat_flows(oracle, weights, golden):
    """(4') convs + resizes pinned independently of the splat restatement."""
    g, gi = golden("model_256x256_t0500"), golden("flows_identity_splat_256x256")
    flow = None
    with torch.no_grad():
        for level in range(5, 0, -1):
            flow = oracle.flow_level(weights, torch.from_numpy(g["feat%d" % level]), flow, splat=lambda a, b: a)
            np.testing.assert_allclose(flow.numpy(), gi["flow%d" % level], atol=1e-5)


def test_ops_match_reference(oracle, weights, golden):
    g = golden("ops")
    T = torch.from_numpy
    with torch.no_grad():
        np.testing.assert_allclose(oracle.bwarp(T(g["bwarp_x"]), T(g["bwarp_flo"])).numpy(), g["bwarp_out"], atol=1e-6)
        np.testing.assert_allclose(oracle.bwarp(T(g["bwarp_x"]), T(g["bwarp_flo"]), False).numpy(),
                                   g["bwarp_out_nomask"], atol=1e-6)
        np.testing.assert_allclose(oracle.refine_unet(weights, T(g["unet_in"])).numpy(), g["unet_out"], atol=1e-5)
        p = oracle.to_pca_diff(T(g["pca_in"]), weights["Mean8"], weights["EV8"], weights["meanVec8"])
        assert p.dtype == torch.float64
        np.testing.assert_allclose(p.numpy(), g["pca_out"], atol=1e-12)
        f = T(g["feat_in"])
        np.testing.assert_allclose(oracle.rec_ctx_ds(weights, f).numpy(), g["rec_ctx_ds_out"], atol=1e-5)
        np.testing.assert_allclose(oracle.conv_flow_bottom(weights, f).numpy(), g["conv_flow_bottom_out"][:, :4], atol=1e-5)
        np.testing.assert_allclose(oracle._conv(weights, "vfinet.conv_flow1", f).numpy(), g["conv_flow1_out"], atol=1e-5)
        np.testing.assert_allclose(oracle.conv_flow2(weights, T(g["flow2_in"])).numpy(), g["conv_flow2_out"], atol=1e-5)


def test_pca_is_8x8_stride8_conv(oracle, weights):
    """F5: the projection equals a per-plane 8x8/stride-8 convolution."""
    g = torch.Generator().manual_seed(3)
    pl = torch.rand(6, 24, 40, generator=g, dtype=torch.float64) * 2 - 1
    EV, M, mv = weights["EV8"], weights["Mean8"], weights["meanVec8"]
    wk = (EV / mv[:, None]).view(16, 1, 8, 8)
    bk = -(EV @ M) / mv
    y = torch.nn.functional.conv2d(pl.unsqueeze(1), wk, bk, stride=8).reshape(96, 3, 5)
    np.testing.assert_allclose(y.numpy(), oracle.pca_project_raw(pl, M, EV, mv).numpy(), atol=1e-12)


# ---- known-answer tests for the CUDA-only operators (no executable reference) ----

def test_splat_zero_flow_identity(oracle):
    x = torch.rand(1, 3, 9, 11) * 2 - 1
    out = oracle.function_softsplat(x, torch.zeros(1, 2, 9, 11), None, "softmax")
    np.testing.assert_allclose(out.numpy(), x.numpy(), atol=1e-6)


def test_splat_integer_shift_and_holes(oracle):
    x = torch.rand(1, 2, 6, 8) * 2 - 1
    flow = torch.zeros(1, 2, 6, 8)
    flow[:, 0] = 2.0
    flow[:, 1] = -1.0
    out = oracle.function_softsplat(x, flow, None, "softmax")
    np.testing.assert_allclose(out[..., :5, 2:].numpy(), x[..., 1:, :6].numpy(), atol=1e-6)
    assert (out[..., :, :2] == -1).all() and (out[..., 5, :] == -1).all()       # holes -> -1 (softSplat.py:346-349)


def test_splat_half_pixel_weights_and_mass(oracle):
    inp = torch.zeros(1, 1, 5, 5)
    inp[0, 0, 2, 2] = 1.0
    flow = torch.full((1, 2, 5, 5), 0.5)
    out = oracle.splat_forward(inp, flow)
    assert out[0, 0, 2:4, 2:4].tolist() == [[0.25, 0.25], [0.25, 0.25]]
    z = torch.randn(1, 1, 7, 9)
    flow = torch.rand(1, 2, 7, 9) * 1.5 - 0.75
    acc = oracle.splat_forward(torch.cat([torch.rand(1, 2, 7, 9) * z.exp(), z.exp()], 1), flow)
    # interior sources (all 4 corners in bounds) deposit exactly e^z
    fx = torch.arange(9.).view(1, 9) + flow[0, 0]
    fy = torch.arange(7.).view(7, 1) + flow[0, 1]
    w_in = 0
    for tx, wx in ((fx.floor(), fx.floor() + 1 - fx), (fx.floor() + 1, fx - fx.floor())):
        for ty, wy in ((fy.floor(), fy.floor() + 1 - fy), (fy.floor() + 1, fy - fy.floor())):
            ok = (tx >= 0) & (tx < 9) & (ty >= 0) & (ty < 7)
            w_in = w_in + wx * wy * ok
    assert math.isclose(acc[0, -1].sum().item(), (z.exp()[0, 0] * w_in).sum().item(), rel_tol=1e-5)


def test_splat_softmax_collision_mix(oracle):
    img = torch.zeros(1, 1, 1, 4)
    img[0, 0, 0, 0], img[0, 0, 0, 2] = -1.0, 1.0
    flow = torch.zeros(1, 2, 1, 4)
    flow[0, 0, 0, 0], flow[0, 0, 0, 2] = 1.0, -1.0        # both land on x=1
    flow[0, 0, 0, 1] = 2.0                                # move the resident of x=1 away
    z = torch.zeros(1, 1, 1, 4)
    z[0, 0, 0, 2] = math.log(3.0)
    out = oracle.function_softsplat(img, flow, z, "softmax")
    assert math.isclose(out[0, 0, 0, 1].item(), (0 * 1 + 1 * 3) / 4 * 2 - 1, abs_tol=1e-6)   # 1:3 mix


def test_splat_modes(oracle):
    x = torch.rand(1, 2, 4, 4)
    f = torch.zeros(1, 2, 4, 4)
    m = torch.rand(1, 1, 4, 4) + 0.5
    np.testing.assert_allclose(oracle.function_softsplat(x, f, None, "summation").numpy(), ((x - 0.5) * 2).numpy(), atol=1e-6)
    np.testing.assert_allclose(oracle.function_softsplat(x, f, None, "average").numpy(), ((x - 0.5) * 2).numpy(), atol=1e-6)
    np.testing.assert_allclose(oracle.function_softsplat(x, f, m, "linear").numpy(), ((x - 0.5) * 2).numpy(), atol=1e-6)
    with pytest.raises(AssertionError):
        oracle.function_softsplat(x, f, torch.rand(1, 2, 4, 4), "softmax")


def test_correlation_known_answers(oracle):
    a = torch.randn(2, 5, 12, 14)
    out = oracle.correlation(a, a)
    assert out.shape == (2, 81, 12, 14)
    np.testing.assert_allclose(out[:, 40].numpy(), (a * a).mean(1).numpy(), atol=1e-6)
    b = torch.roll(a, shifts=(2, -3), dims=(2, 3))        # b[y,x] = a[y-2,x+3]  => a[y,x] = b[y+2,x-3]
    out = oracle.correlation(a, b)
    am = out[:, :, 5:-5, 5:-5].mean((0, 2, 3)).argmax().item()
    assert am == (2 + 4) * 9 + (-3 + 4)
    # zero padding at borders: displacement (-4,-4) at pixel (0,0) reads padding
    assert out[0, 0, 0, 0].item() == 0.0
    # unfold-based restatement
    un = torch.nn.functional.unfold(torch.nn.functional.pad(b, (4, 4, 4, 4)), 9).view(2, 5, 81, 12, 14)
    np.testing.assert_allclose(out.numpy(), (a.unsqueeze(2) * un).mean(1).numpy(), atol=1e-6)


def test_psnr_known_answer(oracle):
    a = np.zeros((4, 4, 3))
    b = np.full((4, 4, 3), 5.0)
    assert math.isclose(oracle.psnr(a, b), 10 * math.log10(255 ** 2 / 25.0))
    assert oracle.psnr(a, a) == float("inf")


def test_oracle_backward_restatements_match_autograd_of_the_forward():
    """The backward kernels of the two custom ops (softSplat.py:54-158, correlation.py:114-242) restated from the kernel text
    must equal the autograd gradients of the forward restatements (the forward ops are what the reference's tests pin)."""
    import fldr_oracle as O
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(2, 3, 13, 17, generator=g) * 2 - 1).requires_grad_(True)
    flow = ((torch.rand(2, 2, 13, 17, generator=g) - 0.5) * 7).requires_grad_(True)
    out = O.splat_forward(x, flow)
    go = torch.randn(out.shape, generator=g)
    gx, gf = torch.autograd.grad(out, [x, flow], go)
    rx, rf = O.splat_backward(x.detach(), flow.detach(), go)
    assert (gx - rx).abs().max() < 2e-6 and (gf - rf).abs().max() < 2e-5
    a = torch.randn(1, 5, 11, 14, generator=g).requires_grad_(True)
    b = torch.randn(1, 5, 11, 14, generator=g).requires_grad_(True)
    c = O.correlation(a, b)
    gc = torch.randn(c.shape, generator=g)
    ga, gb = torch.autograd.grad(c, [a, b], gc)
    ra, rb = O.correlation_backward(a.detach(), b.detach(), gc)
    assert (ga - ra).abs().max() < 2e-6 and (gb - rb).abs().max() < 2e-6


def test_ssim_y_oracle_known_answers(oracle):
    """oracle.ssim_y restates utils.ssim_bgr (utils.py:662-669) + scikit-image 0.19.3's structural_similarity defaults.
    The reference holds no SSIM vectors and scikit-image is not installed here, so it is pinned by known answers:
    identical images -> 1; an INDEPENDENT direct evaluation of the definition (explicit 7x7 windows, no uniform_filter)
    on the interior; monotone in the distortion; and committed values (tests/golden/ssim_y_known_answers.json, generated
    by the oracle itself: a regression pin, not a reference vector)."""
    import json
    import os
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (23, 31, 3)).astype(np.float64)
    b = np.clip(a + rng.normal(0, 12, a.shape), 0, 255).round()
    assert oracle.ssim_y(a, a) == pytest.approx(1.0, abs=1e-15)

    def direct(x, y):                                       # the definition, window by window
        w = np.array([0.256788235294118, 0.504129411764706, 0.097905882352941])
        X = x[:, :, ::-1] @ w + 16.0
        Y = y[:, :, ::-1] @ w + 16.0
        R = Y.max() - Y.min()
        C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
        vals = []
        for i in range(3, X.shape[0] - 3):
            for j in range(3, X.shape[1] - 3):
                p, q = X[i - 3:i + 4, j - 3:j + 4].ravel(), Y[i - 3:i + 4, j - 3:j + 4].ravel()
                ux, uy = p.mean(), q.mean()
                vx, vy = p.var(ddof=1), q.var(ddof=1)
                vxy = ((p - ux) * (q - uy)).sum() / 48.0
                vals.append((2 * ux * uy + C1) * (2 * vxy + C2) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
        return float(np.mean(vals))
    assert oracle.ssim_y(a, b) == pytest.approx(direct(a, b), abs=1e-11)
    worse = np.clip(a + rng.normal(0, 40, a.shape), 0, 255).round()
    assert oracle.ssim_y(a, worse) < oracle.ssim_y(a, b) < 1.0
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim_y_known_answers.json")))
    u = oracle.synthetic_pair(96, 128, seed=3, quadrant=True).numpy()
    t, p = np.transpose(u[0], (1, 2, 0)).astype(np.float64), np.transpose(u[1], (1, 2, 0)).astype(np.float64)
    assert oracle.ssim_y(t, p) == pytest.approx(fx["frame1_vs_frame0"], abs=1e-12)
    assert oracle.ssim_y(t, np.clip(t + 5, 0, 255)) == pytest.approx(fx["plus5"], abs=1e-12)
    assert oracle.ssim_y(t, np.round((t + p) / 2)) == pytest.approx(fx["blur_like_half"], abs=1e-12)
