"""Full-size parity at the X-Test / Xiph geometry (4096x2160 -> padded 2304x4096; main.py:795): BASELINE configs 3
(8x multi-frame, t = 1/8 .. 7/8, pair-invariant cache) and 5 (fp16-input convolutions), through the C ABI, against the
CPU oracle on the same seeded pair.  The 288x512 feature maps have different tile counts / XCD splits than the 3840-wide
case (16 tiles per row, 1152 conv units) in the persistent conv, band splat and stride-2 encoder kernels.
Tolerances are ~10x the errors measured on MI355X, not looser."""
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W = 2160, 4096


@pytest.fixture(scope="module")
def hip():
    import fldr_hip
    fldr_hip.lib()
    return fldr_hip


@pytest.fixture(scope="module")
def model(dev):
    import fldr_harness as Hn
    m, _, a = Hn.prepare_model(dev)
    return m, a


@pytest.fixture(scope="module")
def xtest(dev, oracle, weights):
    """One synthetic 4096x2160 pair with per-quadrant motion (occlusions / holes) + the oracle's frames at t = 0.5 and
    t = 0.125 (two CPU forwards of ~10 s each on the GPU box)."""
    import fldr_harness as Hn
    frames = Hn.frames_from_uint8(Hn.synthetic_pair(H, W, seed=2, quadrant=True))
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    pyr = oracle.pad_and_pyramid(frames)
    assert tuple(pyr[0].shape[-2:]) == (2304, 4096)
    refs = {}
    with torch.no_grad():
        for tv in (0.5, 0.125):
            refs[tv] = oracle.forward(weights, pyr, torch.tensor([[tv]]))[:, :, :H, :W]
    return frames.to(dev), refs


def _errs(out, ref):
    import fldr_harness as Hn
    err = (out.double().cpu() - ref.double()).abs()
    return err.max().item(), err.mean().item(), Hn.psnr(Hn.to_uint8_image(ref[0]), Hn.to_uint8_image(out[0]))


@pytest.mark.timeout(900)
def test_xtest_geometry_forward_matches_oracle(hip, dev, model, xtest):
    """Config 3 geometry, t = 0.5: whole forward vs the oracle (fp32-class bounds: max 1e-4, mean 1e-6, >= 90 dB)."""
    import fldr_harness as Hn
    m, a = model
    frames, refs = xtest
    out = Hn.interpolate(m, a, frames, torch.tensor([[0.5]], device=dev))
    assert out.shape == (1, 3, H, W) and out.dtype == torch.float64 and torch.isfinite(out).all()
    mx, mean, p = _errs(out, refs[0.5])
    print("4096x2160 t=0.5: max|err| %.2e mean %.2e PSNR(8-bit) %.1f dB" % (mx, mean, p))
    assert mx <= 1e-4 and mean <= 1e-6 and p >= 90.0


@pytest.mark.timeout(900)
def test_xtest_geometry_multi_t_pair_cache(hip, dev, model, xtest):
    """Config 3: the 7 outputs of a pair (main.py:833-867) with the pair-invariant stage computed once == 7 independent
    forwards BIT FOR BIT (the default path is deterministic since the splats accumulate in fp64 LDS tiles; the reference's
    own GPU output is run-to-run non-deterministic) and == the oracle at t = 1/8 and 1/2."""
    import fldr_harness as Hn
    m, a = model
    frames, refs = xtest
    ts = [k / 8 for k in range(1, 8)]
    cached = Hn.interpolate_multi(m, a, frames, ts)
    assert m.pair_cache is False and m._pair_state is None and len(cached) == 7
    for tv, c in zip(ts, cached):
        plain = Hn.interpolate(m, a, frames, torch.tensor([[tv]], device=dev))
        assert torch.equal(c, plain), "cached vs uncached at t=%g: %.2e" % (tv, (c - plain).abs().max().item())
        if tv in refs:
            mx, mean, p = _errs(c, refs[tv])
            print("4096x2160 multi-t t=%g: max|err| %.2e mean %.2e PSNR(8-bit) %.1f dB" % (tv, mx, mean, p))
            assert mx <= 1e-4 and mean <= 1e-6 and p >= 90.0


@pytest.mark.timeout(900)
def test_xiph_geometry_fp16_conv_mode(hip, dev, model, xtest):
    """Config 5 at its real size: 3x3 convolutions with plain fp16 inputs (fp32 accumulation).  Not fp32-class by design:
    the PSNR of its rounded 8-bit frame against the oracle's is reported; >= 65 dB (measured 74 dB at 512x768), i.e. an
    error four orders of magnitude below the 0.02 dB budget of north_star."""
    import fldr_harness as Hn
    m, a = model
    frames, refs = xtest
    prev = hip.CONV_PRECISION
    try:
        hip.CONV_PRECISION = "fp16"
        out = Hn.interpolate(m, a, frames, torch.tensor([[0.5]], device=dev))
    finally:
        hip.CONV_PRECISION = prev
    mx, mean, p = _errs(out, refs[0.5])
    print("4096x2160 fp16-input convs: max|err| %.2e mean %.2e PSNR(8-bit vs oracle) %.1f dB" % (mx, mean, p))
    assert torch.isfinite(out).all() and p >= 65.0 and mean <= 2e-4


def test_deterministic_mode_is_bitwise_reproducible(hip, dev, model, xtest):
    """FLDR_SPLAT_FEATURES=gather (the atomic-free feature splat, kept as an opt-in mode; the default fp64-LDS-atomic splat is
    deterministic too, asserted in test_4k_forward_properties / the multi-t tests): two forwards are bit-identical and within
    the usual bounds of the oracle."""
    import fldr_harness as Hn
    m, a = model
    frames, refs = xtest
    prev = hip.SPLAT_FEATURES
    try:
        hip.SPLAT_FEATURES = "gather"
        t = torch.tensor([[0.5]], device=dev)
        o1 = Hn.interpolate(m, a, frames, t)
        o2 = Hn.interpolate(m, a, frames, t)
    finally:
        hip.SPLAT_FEATURES = prev
    assert torch.equal(o1, o2)
    mx, mean, p = _errs(o1, refs[0.5])
    print("4096x2160 deterministic mode: max|err| %.2e mean %.2e PSNR(8-bit) %.1f dB" % (mx, mean, p))
    assert mx <= 1e-4 and mean <= 1e-6 and p >= 90.0


@pytest.mark.timeout(900)
def test_4k_strong_nonrigid_motion_matches_oracle(hip, dev, model, oracle, weights):
    """Whole forward at 3840x2160 on a pair under a strong smoothly varying motion field (2.5 % zoom + 0.8 degree rotation +
    shift: displacements up to ~85 px that change from pixel to pixel, large occluded / disoccluded borders) — the flows the
    scatter kernels find hardest, through the model rather than through operator tests — against the oracle.

    The parity configuration (fp32 PCA residual, the default) meets the bound of every other whole-frame test: the backward warp's
    hard mask threshold (fLDRnet.py:573-574) may flip on isolated pixels — at most one value in a million beyond 1e-4 — mean error
    <= 1e-6, >= 90 dB.  No allowance is derived from the oracle's conditioning report; it is printed for the record only.

    FLDR_PCA_F32=0 (features split-packed only; opt-in, NOT the parity configuration) is run on the same pair and reported: its
    2.4e-7 feature difference flips one nearly empty target cell of a feature splat at the frame border on this pair (a 24 x 46 px
    patch, 4.9e-5 of the values beyond 1e-4, 91.7 dB).  Its assertions: the differences beyond 1e-4 stay inside ONE bounded patch
    (<= 64 x 64 px) — anything outside it fails, and so does any such difference when the oracle reports no ill-conditioned cell — and
    mean error / PSNR as above."""
    import fldr_harness as Hn
    m, a = model
    Hs, Ws = 2160, 3840
    frames = Hn.frames_from_uint8(Hn.synthetic_pair_varying(Hs, Ws, seed=31, zoom=1.025, rot_deg=0.8, shift=(7.0, -4.0)))
    t = torch.tensor([[0.375]])
    assert hip.PCA_F32, "the parity configuration adds the fp32 PCA features in rec_ctx_ds.2"
    out = Hn.interpolate(m, a, frames.to(dev), t.to(dev))
    hip.check_range()
    try:
        hip.PCA_F32 = False
        out_packed = Hn.interpolate(m, a, frames.to(dev), t.to(dev))
    finally:
        hip.PCA_F32 = True
    hip.check_range()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    keep = {}
    with torch.no_grad():
        ref = oracle.forward(weights, oracle.pad_and_pyramid(frames), t, keep=keep, conditioning=True)[:, :, :Hs, :Ws]
    n_ill = sum(keep["ill_conditioned_splat_cells"])
    err = (out.double().cpu() - ref.double()).abs()
    frac = (err > 1e-4).double().mean().item()
    p = Hn.psnr(Hn.to_uint8_image(ref[0]), Hn.to_uint8_image(out[0]))
    print("3840x2160 strong non-rigid motion: max|err| %.2e mean %.2e, %.2e of the values beyond 1e-4, PSNR(8-bit) %.1f dB; "
          "ill-conditioned feature-splat cells per level (oracle, eps %.0e, +-%.0e px; not used by the bound): %s"
          % (err.max().item(), err.mean().item(), frac, p, oracle.SPLAT_COND_EPS, oracle.SPLAT_COND_DELTA, keep["ill_conditioned_splat_cells"]))
    assert frac <= 1e-6 and err.mean().item() <= 1e-6 and p >= 90.0
    # --- the opt-in packed-residual mode
    errp = (out_packed.double().cpu() - ref.double()).abs()
    badp = (errp > 1e-4).any(1)[0]                                        # [H, W]
    fracp = (errp > 1e-4).double().mean().item()
    pp = Hn.psnr(Hn.to_uint8_image(ref[0]), Hn.to_uint8_image(out_packed[0]))
    ys, xs = torch.nonzero(badp, as_tuple=True)
    box = (int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())) if len(ys) else None
    print("  FLDR_PCA_F32=0 (opt-in): max|err| %.2e mean %.2e, %.2e of the values beyond 1e-4, PSNR(8-bit) %.1f dB, bounding box of the "
          "pixels beyond 1e-4 (y0, y1, x0, x1): %s" % (errp.max().item(), errp.mean().item(), fracp, pp, box))
    assert errp.mean().item() <= 1e-6 and pp >= 90.0
    if fracp > 1e-6:
        assert n_ill >= 1, "differences beyond 1e-4 without any ill-conditioned splat cell"
        assert box[1] - box[0] < 64 and box[3] - box[2] < 64, "differences beyond 1e-4 outside one 64 x 64 patch: %s" % (box,)
