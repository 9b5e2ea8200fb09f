"""CPU oracle for the fLDRnet per-frame-pair inference path.

TEST INFRASTRUCTURE ONLY.  This file restates, on the CPU, the algorithm of the
reference (visinf/fldr-vfi) for the hot path named in BASELINE.json.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product (`fldr-vfi_amd/`) never does and has no CPU fallback.

Arithmetic that the reference delegates to PyTorch (convolutions,
`F.interpolate`, `F.grid_sample`, softmax) is delegated to PyTorch-CPU here as
well; everything the reference implements itself (the CUDA splat and
correlation kernels, the PCA block projection, the level driver) is restated
from the reference text.  All `file:line` citations are relative to the
reference repository root.

Pinning: every function below is checked in `tests/test_oracle_golden.py`
against golden vectors produced by importing the reference's own model code in
the build container (`tools/make_golden.py`).  The two CUDA-only operators
(softmax splat, cost-volume correlation) have no executable reference here
(softSplat.py:251-252, correlation.py:343-344 raise NotImplementedError on
CPU): for those two ops parity is pinned by known-answer tests derived from the
kernel text, not by reference outputs.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# softmax splatting  (softSplat.py)
# --------------------------------------------------------------------------

def splat_forward(inp, flow):
    """Summation forward splat, kernel_Softsplat_updateOutput (softSplat.py:12-52).

    inp [N,C,H,W] fp32, flow [N,2,H,W] fp32 (channel 0 = x, 1 = y, pixels).
    The reference accumulates fp32 products with fp32 atomicAdd in arbitrary
    order; here the fp32 products are accumulated in fp64 and rounded once, so
    the result is within one fp32 rounding of every accumulation order.
    """
    N, C, H, W = inp.shape
    assert flow.shape == (N, 2, H, W)
    inp = inp.float()
    flow = flow.float()
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(N, H, W)
    ys = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(N, H, W)
    fx = xs + flow[:, 0]          # softSplat.py:23
    fy = ys + flow[:, 1]          # softSplat.py:24
    assert torch.isfinite(fx).all() and torch.isfinite(fy).all()  # :25-26
    x0 = torch.floor(fx)
    y0 = torch.floor(fy)
    x1 = x0 + 1
    y1 = y0 + 1
    # softSplat.py:35-38 (fp32 arithmetic)
    w_nw = (x1 - fx) * (y1 - fy)
    w_ne = (fx - x0) * (y1 - fy)
    w_sw = (x1 - fx) * (fy - y0)
    w_se = (fx - x0) * (fy - y0)
    out = torch.zeros(N, C, H * W, dtype=torch.float64)
    for (tx, ty, w) in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        ok = (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)      # softSplat.py:39-49
        idx = (ty.clamp(0, H - 1) * W + tx.clamp(0, W - 1)).long()
        val = (inp * w.unsqueeze(1)).double() * ok.unsqueeze(1)  # fp32 product, fp64 sum
        for n in range(N):
            out[n].index_add_(1, idx[n].reshape(-1), val[n].reshape(C, -1))
    return out.float().view(N, C, H, W)


def splat_backward(inp, flow, grad_out):
    """_FunctionSoftsplat.backward: kernel_Softsplat_updateGradInput (softSplat.py:54-100) and
    kernel_Softsplat_updateGradFlow (:102-158), restated from the kernel text.  -> (gradInput, gradFlow).
    fp32 products in the reference's factorisation, accumulated in fp64 over corners / channels and rounded once."""
    N, C, H, W = inp.shape
    inp, flow, g = inp.float(), flow.float(), grad_out.float()
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(N, H, W)
    ys = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(N, H, W)
    fx, fy = xs + flow[:, 0], ys + flow[:, 1]                         # :67-68, :115-116
    x0, y0 = torch.floor(fx), torch.floor(fy)
    x1, y1 = x0 + 1, y0 + 1
    w = {"nw": (x1 - fx) * (y1 - fy), "ne": (fx - x0) * (y1 - fy), "sw": (x1 - fx) * (fy - y0), "se": (fx - x0) * (fy - y0)}   # :79-82
    dx = {"nw": -1.0 * (y1 - fy), "ne": +1.0 * (y1 - fy), "sw": -1.0 * (fy - y0), "se": +1.0 * (fy - y0)}                 # :131-135
    dy = {"nw": (x1 - fx) * -1.0, "ne": (fx - x0) * -1.0, "sw": (x1 - fx) * +1.0, "se": (fx - x0) * +1.0}                 # :136-141
    pos = {"nw": (x0, y0), "ne": (x1, y0), "sw": (x0, y1), "se": (x1, y1)}
    gin = torch.zeros(N, C, H, W, dtype=torch.float64)
    gfx = torch.zeros(N, H, W, dtype=torch.float64)
    gfy = torch.zeros(N, H, W, dtype=torch.float64)
    gflat = g.reshape(N, C, H * W)
    for k in ("nw", "ne", "sw", "se"):
        tx, ty = pos[k]
        ok = (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)                # :83-94, :144-155
        idx = (ty.clamp(0, H - 1) * W + tx.clamp(0, W - 1)).long().view(N, 1, H * W).expand(N, C, H * W)
        gk = torch.gather(gflat, 2, idx).view(N, C, H, W) * ok.unsqueeze(1)
        gin += (gk * w[k].unsqueeze(1)).double()
        gfx += ((inp * gk) * dx[k].unsqueeze(1)).double().sum(1)
        gfy += ((inp * gk) * dy[k].unsqueeze(1)).double().sum(1)
    return gin.float(), torch.stack([gfx, gfy], 1).float()


def correlation_backward(first, second, grad_out):
    """_FunctionCorrelation.backward: kernel_Correlation_updateGradFirst (correlation.py:114-168) and
    kernel_Correlation_updateGradSecond (:170-242) for stride 1 / kernel 1 / pad 4, restated from the kernel text.
    -> (gradFirst, gradSecond)."""
    N, C, H, W = first.shape
    g = grad_out.double()
    f, sp = first.double(), F.pad(second.double(), (4, 4, 4, 4))
    gpad = F.pad(g, (4, 4, 4, 4))
    fpad = F.pad(f, (4, 4, 4, 4))
    g1 = torch.zeros(N, C, H, W, dtype=torch.float64)
    g2 = torch.zeros(N, C, H, W, dtype=torch.float64)
    for p in range(-4, 5):
        for o in range(-4, 5):
            op = (p + 4) * 9 + (o + 4)
            # gradFirst[y,x] += gradOut[op,y,x] * second[y+p,x+o]                       (:150-163)
            g1 += g[:, op:op + 1] * sp[:, :, 4 + p:4 + p + H, 4 + o:4 + o + W]
            # gradSecond[y,x] += gradOut[op,y-p,x-o] * first[y-p,x-o] where in range     (:205-235)
            g2 += gpad[:, op:op + 1, 4 - p:4 - p + H, 4 - o:4 - o + W] * fpad[:, :, 4 - p:4 - p + H, 4 - o:4 - o + W]
    return (g1 / C).float(), (g2 / C).float()                                        # :166, :240


def function_softsplat(img, flow, metric, mode="softmax"):
    """FunctionSoftsplat (softSplat.py:320-352), including its quirks:
    the (x+1)/2 pre-scale happens only for 'softmax' (:334) while the
    (x-0.5)*2 post-scale happens for every mode (:349)."""
    assert metric is None or metric.shape[1] == 1
    assert mode in ("summation", "average", "linear", "softmax")
    if mode == "average":
        inp = torch.cat([img, img.new_ones(img.shape[0], 1, img.shape[2], img.shape[3])], 1)
    elif mode == "linear":
        inp = torch.cat([img * metric, metric], 1)
    elif mode == "softmax":
        x = (img + 1) / 2
        if metric is None:
            inp = torch.cat([x * 1, torch.ones_like(x[:, :1])], 1)
        else:
            e = metric.exp()
            inp = torch.cat([x * e, e], 1)
    else:
        inp = img
    out = splat_forward(inp, flow)
    if mode != "summation":
        norm = out[:, -1:, :, :].clone()
        norm[norm == 0.0] = 1.0
        out = out[:, :-1, :, :] / norm
    return (out - 0.5) * 2


SPLAT_COND_EPS = 1e-4       # a target cell whose normaliser stays below this is "nearly empty" ...
SPLAT_COND_DELTA = 1e-3     # ... under flow perturbations of this many pixels


def splat_ill_conditioned_cells(flow, eps=SPLAT_COND_EPS, delta=SPLAT_COND_DELTA):
    """Test diagnostic (not part of the reference): target cells of a softmax splat WITHOUT metric whose value is ill-conditioned
    in the flow.  The splat divides by the cell's normaliser n = sum of the bilinear weights that reach it and writes a hole where
    n == 0 (softSplat.py:343-349): a cell with a tiny n — a single source whose footprint grazes it — flips between "hole" and
    "that source's full value" when the flow moves by one ulp, which is what the reference's own fp32 atomics and any fp32
    re-association of the convolutions in front of it do from run to run.  Returns the number of cells whose normaliser,
    over the flow and its four axis perturbations by `delta` px, is > 0 at least once and < eps at least once."""
    ones = torch.ones_like(flow[:, :1])
    n_min = n_max = None
    for dx, dy in ((0.0, 0.0), (delta, 0.0), (-delta, 0.0), (0.0, delta), (0.0, -delta)):
        f = flow.clone()
        f[:, 0] += dx
        f[:, 1] += dy
        n = splat_forward(ones, f)
        n_min = n if n_min is None else torch.minimum(n_min, n)
        n_max = n if n_max is None else torch.maximum(n_max, n)
    return int(((n_min < eps) & (n_max > 0)).sum())


# --------------------------------------------------------------------------
# cost volume  (OpticalFlow/correlation.py)
# --------------------------------------------------------------------------

def correlation(first, second):
    """FunctionCorrelation forward (correlation.py:294-348, kernels :17-112):
    out[n,(dy+4)*9+(dx+4),y,x] = mean_c first[n,c,y,x]*second[n,c,y+dy,x+dx],
    zero padding 4 (rbot* are new_zeros, :297-298)."""
    N, C, H, W = first.shape
    pad = F.pad(second.double(), (4, 4, 4, 4))
    f = first.double()
    out = torch.empty(N, 81, H, W, dtype=torch.float64)
    for p in range(-4, 5):            # s2p = top_channel / 9 - 4  (:82)
        for o in range(-4, 5):        # s2o = top_channel % 9 - 4  (:81)
            win = pad[:, :, 4 + p:4 + p + H, 4 + o:4 + o + W]
            out[:, (p + 4) * 9 + (o + 4)] = (f * win).sum(1) / C   # :108
    return out.float()


# --------------------------------------------------------------------------
# PCA block projection  (pca_comp.py:473-528)
# --------------------------------------------------------------------------

def pca_project_raw(planes, mean, EV, mean_vec):
    """Un-normalised projection: y[p*K+k,by,bx] (fp64), pca_comp.py:489-518."""
    P, H, W = planes.shape
    if H % 8 or W % 8:
        raise Exception("in to_pca_diff the image is not padded right." + str(H) + " " + str(W))
    K = EV.shape[0]
    blk = planes.double().view(P, H // 8, 8, W // 8, 8).permute(0, 1, 3, 2, 4).reshape(P, H // 8, W // 8, 64)
    y = torch.matmul(blk - mean.double(), EV.double().t()) / mean_vec.double()  # :502-511
    return y.permute(0, 3, 1, 2).reshape(P * K, H // 8, W // 8)                  # :516-518


def to_pca_diff(planes, mean, EV, mean_vec):
    """to_pca_diff (pca_comp.py:473-528): projection + GLOBAL min/max rescale to [-1,1], fp64."""
    y = pca_project_raw(planes, mean, EV, mean_vec)
    mi, ma = y.min(), y.max()                      # :521-522
    return ((y - mi) / (ma - mi)) * 2 - 1          # :523-526


# --------------------------------------------------------------------------
# backward warp  (fLDRnet.py:546-581)
# --------------------------------------------------------------------------

def bwarp(x, flo, withmask=True):
    B, C, H, W = x.shape
    xx = torch.arange(0, W).view(1, 1, 1, W).expand(B, 1, H, W)
    yy = torch.arange(0, H).view(1, 1, H, 1).expand(B, 1, H, W)
    vgrid = torch.cat((xx, yy), 1).float() + flo
    gx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0   # :564
    gy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0   # :565
    g = torch.stack((gx, gy), dim=3)
    out = F.grid_sample(x, g, align_corners=False)  # :568 (default mode/padding)
    if not withmask:
        return out
    mask = F.grid_sample(torch.ones_like(x), g, align_corners=False)  # :569-570
    mask = mask.masked_fill(mask < 0.999, 0)
    mask = mask.masked_fill(mask > 0, 1)
    return out * mask


BWARP_MASK_THRESHOLD = 0.999   # fLDRnet.py:573


def bwarp_mask_value(shape, flo):
    """Test diagnostic (not part of the reference): the value the hard threshold of bwarp is applied to (fLDRnet.py:569-574),
    [B,1,H,W].  A pixel whose value lies within a rounding error of 0.999 is ill-conditioned: the reference's own grid_sample
    flips it between "kept" and "zeroed" with any re-association of the grid arithmetic; tests confine differences to such pixels."""
    B, _, H, W = shape
    xx = torch.arange(0, W).view(1, 1, 1, W).expand(B, 1, H, W)
    yy = torch.arange(0, H).view(1, 1, H, 1).expand(B, 1, H, W)
    vgrid = torch.cat((xx, yy), 1).float() + flo
    gx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
    gy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
    return F.grid_sample(torch.ones(B, 1, H, W), torch.stack((gx, gy), dim=3), align_corners=False)


# --------------------------------------------------------------------------
# network pieces  (fLDRnet.py)
# --------------------------------------------------------------------------

def _conv(w, name, x, stride=1, pad=1):
    return F.conv2d(x, w[name + ".weight"], w[name + ".bias"], stride=stride, padding=pad)


def rec_ctx_ds(w, x):
    """rec_ctx_ds + residual (fLDRnet.py:44-49,162)."""
    y = F.relu(_conv(w, "rec_ctx_ds.0", x))
    y = F.relu(_conv(w, "rec_ctx_ds.2", y))
    return y + x


def conv_flow_bottom(w, x):
    """fLDRnet.py:318-330, 379-380."""
    for i in (0, 2, 4, 6):
        x = F.relu(_conv(w, "vfinet.conv_flow_bottom.%d" % i, x))
    return _conv(w, "vfinet.conv_flow_bottom.8", x)[:, :4]


def conv_flow2(w, x):
    """fLDRnet.py:335-345."""
    for i in (0, 2, 4, 6):
        x = F.relu(_conv(w, "vfinet.conv_flow2.%d" % i, x))
    return _conv(w, "vfinet.conv_flow2.8", x)


def refine_unet(w, x):
    """PCARefineUNet.forward (fLDRnet.py:619-644)."""
    p = "vfinet.refine_unet."
    enc1 = F.relu(_conv(w, p + "enc1", x, 2, 1))
    enc2 = F.relu(_conv(w, p + "enc2", enc1, 2, 1))
    out = F.relu(_conv(w, p + "enc3", enc2, 2, 1))
    out = F.relu(_conv(w, p + "dec0", out))
    out = F.interpolate(out, scale_factor=2, mode="nearest")
    out = F.relu(_conv(w, p + "dec1", torch.cat((out, enc2), 1)))
    out = F.interpolate(out, scale_factor=2, mode="nearest")
    out = F.relu(_conv(w, p + "dec2", torch.cat((out, enc1), 1)))
    out = F.interpolate(out, scale_factor=2, mode="nearest")
    return _conv(w, p + "dec3", out)


def flow_level(w, feat, flow_prev, splat=None, cond=None):
    """Flow estimation part of DCTVFInet.forward (fLDRnet.py:368-391).
    `splat(img, flow)` defaults to the softmax splat without metric.  cond: a list that receives the number of ill-conditioned
    target cells of this level's two feature splats (splat_ill_conditioned_cells; test diagnostic)."""
    if splat is None:
        splat = lambda a, b: function_softsplat(a, b, None, "softmax")
    B, C, H, W = feat.shape
    feat0, feat1 = feat[:, :48], feat[:, 48:]       # :368-370 (F4 channel split)
    if flow_prev is None:
        return conv_flow_bottom(w, torch.cat((feat0, feat1), 1))
    up = F.interpolate(flow_prev, size=(H, W), mode="bilinear", align_corners=False)  # :384
    up = up * (up.shape[3] / flow_prev.shape[3])                                       # :385
    if cond is not None:
        cond.append(splat_ill_conditioned_cells(up[:, :2]) + splat_ill_conditioned_cells(up[:, 2:]))
    w1 = splat(feat1, up[:, :2])                                                      # :386
    w0 = splat(feat0, up[:, 2:])                                                      # :387
    a = _conv(w, "vfinet.conv_flow1", torch.cat([feat0, w1], 1))
    b = _conv(w, "vfinet.conv_flow1", torch.cat([feat1, w0], 1))
    return conv_flow2(w, torch.cat([a, b, up], 1))[:, :4] + up                          # :389-391


def synthesis_level0(w, flow_l, x_l, t, splat_fn=None, keep=None):
    """Level-0 tail of DCTVFInet.forward (fLDRnet.py:400-524).
    flow_l [B,4,h,w]; x_l [B,3,2,H,W] fp32; t [B,1,1,1] fp32.  Returns fp64 out."""
    if splat_fn is None:
        splat_fn = function_softsplat
    keep = {} if keep is None else keep
    I0, I1 = x_l[:, :, 0], x_l[:, :, 1]
    flow_10, flow_01 = flow_l[:, :2], flow_l[:, 2:]           # :400-401
    flow_t0 = t * flow_01                                      # :404
    flow_t1 = (1 - t) * flow_10                                # :405
    up = x_l.shape[3] / flow_t0.shape[2]                       # :410
    if not float(up).is_integer():
        raise Exception("upscale factor is no integer!!! Upscale factor: " + str(up))
    up = int(up)
    U = lambda f: up * F.interpolate(f, scale_factor=(up, up), mode="bilinear", align_corners=False)
    flow_t0, flow_t1, flow_10, flow_01 = U(flow_t0), U(flow_t1), U(flow_10), U(flow_01)  # :419-422
    za = w["vfinet.z_alpha"]                                   # fp64 (2,), 0-dim picks stay fp32 (F3)
    z0 = torch.mean(float(za[0]) * torch.abs(I0 - bwarp(I1, flow_01)), dim=1, keepdim=True)  # :442-443
    z1 = torch.mean(float(za[1]) * torch.abs(I1 - bwarp(I0, flow_10)), dim=1, keepdim=True)  # :445-446
    warped0 = splat_fn(I0, flow_t0, z0, "softmax")             # :449
    warped1 = splat_fn(I1, flow_t1, z1, "softmax")             # :450
    flowback_0 = bwarp(flow_10 * t, (1 - t) * flow_01)         # :474
    flowback_1 = bwarp(flow_01 * (1 - t), t * flow_10)         # :475
    im0_tot = bwarp(I0, flowback_0)                            # :478
    im1_tot = bwarp(I1, flowback_1)                            # :479
    cat = torch.cat([I0, I1, warped0, warped1, flow_t0, flow_t1, flowback_0, flowback_1, im0_tot, im1_tot], 1)  # :480
    refine_out = refine_unet(w, cat)                           # :501
    T = w["vfinet.T_param"].double()                           # 1-D fp64 -> fp64 tail (F3)
    occ = F.softmax(refine_out[:, 0:6] / T, dim=1)             # :511
    wk = [(1 - t), t, (1 - t), t, (1 - t), t]
    cand = [warped0, warped1, im0_tot, im1_tot, I0, I1]
    divisor = sum(wk[k] * occ[:, k:k + 1] for k in range(4))   # :517
    out = wk[0] * occ[:, 0:1] * cand[0] + wk[1] * occ[:, 1:2] * cand[1]   # :518
    out = out + (wk[2] * occ[:, 2:3] * cand[2] + wk[3] * occ[:, 3:4] * cand[3])   # :520
    out = out + (wk[4] * occ[:, 4:5] * cand[4] + wk[5] * occ[:, 5:6] * cand[5])   # :521
    divisor = divisor + (wk[4] * occ[:, 4:5] + wk[5] * occ[:, 5:6])               # :522
    out = out / divisor                                        # :524
    keep.update(dict(flow_t0=flow_t0, flow_t1=flow_t1, flow_10=flow_10, flow_01=flow_01, z0=z0, z1=z1,
                     warped0=warped0, warped1=warped1, flowback_0=flowback_0, flowback_1=flowback_1,
                     im0_tot=im0_tot, im1_tot=im1_tot, refine_out=refine_out, occ=occ, out=out))
    return out


def forward(w, pyramid, t_value, n_levels=None, identity_splat=False, keep=None,
            out_size=(2160, 4096), conditioning=False):
    """DCTXVFInet.forward, test branch (fLDRnet.py:106-223).

    pyramid: list of [B,3,2,H/2^i,W/2^i] fp32 (B must be 1 for parity);
    t_value [B,1] fp32.  Returns fp64 [B,3,min(H,2160),min(W,4096)]."""
    n_levels = len(pyramid) if n_levels is None else n_levels
    keep = {} if keep is None else keep
    B = pyramid[0].shape[0]
    t = t_value.view(B, 1, 1, 1)
    feats, pcas = [], []
    for i in range(n_levels):
        x = pyramid[i]
        _, _, _, h, wd = x.shape
        p = to_pca_diff(x.reshape(B * 6, h, wd), w["Mean8"], w["EV8"], w["meanVec8"])   # :146
        p = p.reshape(B, 96, h // 8, wd // 8).float()
        pcas.append(p)
        feats.append(rec_ctx_ds(w, p))                                                    # :162
    splat = (lambda a, b: a) if identity_splat else None
    flow = None
    flows = {}
    cond = [] if conditioning else None                       # keep["ill_conditioned_splat_cells"]: per level (coarse to fine, from the second)
    for level in range(n_levels - 1, 0, -1):                                              # :210
        flow = flow_level(w, feats[level], flow, splat, cond)
        flows[level] = flow
    flow = flow_level(w, feats[0], flow, splat, cond)                                     # :218
    flows[0] = flow
    keep.update(dict(pca=pcas, feat=feats, flows=flows))
    if conditioning:
        keep["ill_conditioned_splat_cells"] = cond
    sp = (lambda img, fl, z, mode: img) if identity_splat else None
    out = synthesis_level0(w, flow, pyramid[0], t, sp, keep)
    return out[:, :, :out_size[0], :out_size[1]]                                          # :222


# --------------------------------------------------------------------------
# callers either side of the path (main.py test(), run_on_your_images.py)
# --------------------------------------------------------------------------

def pad_and_pyramid(frames, n_levels=6):
    """frames [B,3,2,H,W] fp32 in [-1,1] -> list of n_levels tensors
    (main.py:840-856 / run_on_your_images.py:124-145): reflect pad right/bottom
    to a multiple of 2^S_tst*8, then direct bicubic downscale by 2^-i."""
    B, C, T, H, W = frames.shape
    div = (2 ** (n_levels - 1)) * 8
    ph = (div - H % div) % div
    pw = (div - W % div) % div
    x = F.pad(frames.reshape(B, C * T, H, W), (0, pw, 0, ph), "reflect").reshape(B, C, T, H + ph, W + pw)
    H2, W2 = H + ph, W + pw
    pyr = [x]
    flat = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H2, W2)
    for i in range(1, n_levels):
        s = 1.0 / (2 ** i)
        d = F.interpolate(flat, scale_factor=s, mode="bicubic", align_corners=False)
        pyr.append(d.reshape(B, T, C, int(H2 * s), int(W2 * s)).permute(0, 2, 1, 3, 4))
    return pyr


def to_uint8_image(pred, OH, OW):
    """main.py:885-894: crop, denorm255, round (values kept as float)."""
    p = np.squeeze(np.asarray(pred))[:, :OH, :OW]
    return np.around(((np.transpose(p, [1, 2, 0]) + 1.0) / 2.0).clip(0.0, 1.0) * 255.0)


def psnr(img_true, img_pred):
    """utils.py:644-652 with XVFIPSNR False: skimage PSNR, data_range=255."""
    err = np.mean((np.asarray(img_true, dtype=np.float64) - np.asarray(img_pred, dtype=np.float64)) ** 2)
    return float("inf") if err == 0 else 10 * math.log10(255.0 ** 2 / err)


_YCBCR_Y = (0.256788235294118, 0.504129411764706, 0.097905882352941)      # utils.py:695 (first row of T), offset 16


def ssim_y(img_true, img_pred):
    """utils.ssim_bgr (utils.py:662-669): SSIM of the Y channel of two [H,W,3] BGR images holding rounded values in
    [0,255].  `to_uint8(x, 0, 255)` (utils.py:637-641) re-rounds in float32, `[:, :, ::-1]` makes it RGB, `_rgb2ycbcr`
    (:690-711) gives Y = T[0] . (R,G,B) + 16 in float64, and `structural_similarity(Y_true, Y_pred, data_range =
    Y_pred.max() - Y_pred.min())` is scikit-image 0.19.3 (requirements.txt:4; NOT installed here) with its defaults,
    restated from its published algorithm (Wang et al. 2004 as implemented in skimage.metrics._structural_similarity):
    win_size 7, uniform window via scipy.ndimage.uniform_filter (the same scipy routine skimage calls), sample covariance
    (NP / (NP - 1)), K1 = 0.01, K2 = 0.03, mean of the SSIM map cropped by (win_size - 1) // 2 = 3 pixels per side.
    Parity unpinned by reference vectors (the reference holds none for SSIM); pinned by known-answer tests."""
    from scipy.ndimage import uniform_filter

    def y_of(img):
        x = np.asarray(img).astype("float32")
        x = np.clip(np.round((x - 0) / (255 - 0) * 255), 0, 255)[:, :, ::-1]        # to_uint8, BGR -> RGB
        t = x.reshape(-1, 3).astype(np.float64) @ np.asarray(_YCBCR_Y, dtype=np.float64)
        return (t + 16.0).reshape(x.shape[0], x.shape[1])

    X, Y = y_of(img_true), y_of(img_pred)
    R = Y.max() - Y.min()
    win, NP = 7, 49
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(X, size=win), uniform_filter(Y, size=win)
    uxx, uyy, uxy = uniform_filter(X * X, size=win), uniform_filter(Y * Y, size=win), uniform_filter(X * Y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return float(S[pad:-pad, pad:-pad].mean(dtype=np.float64))


def synthetic_pair(H, W, seed=0, quadrant=False, device="cpu"):
    """Seeded synthetic uint8 frame pair (bench.py, tests, golden fixtures): a multi-octave (1/f-like) random
    texture, so that every pyramid level sees structure as in natural video; I1 is I0 shifted by (6,4) px, or by
    (+-12,+-8) px per quadrant to force occlusions/holes.  (5x5-smoothed white noise, the first recipe, has no
    content left below 1/8 resolution and the flow network then predicts meaningless +-40 px flows.)"""
    g = torch.Generator().manual_seed(seed)
    Hb, Wb = H + 64, W + 64
    base = torch.zeros(1, 3, Hb, Wb)
    for o in range(8):
        s = 2 ** o
        n = torch.rand(1, 3, -(-Hb // s) + 2, -(-Wb // s) + 2, generator=g)
        if o:
            n = F.interpolate(n, scale_factor=s, mode="bilinear", align_corners=False)
        base += n[..., :Hb, :Wb] * (1.5 ** o)
    base = (base - base.amin()) / (base.amax() - base.amin())
    I0 = base[..., 32:H + 32, 32:W + 32]
    if not quadrant:
        I1 = base[..., 36:H + 36, 38:W + 38]
    else:
        I1 = I0.clone()
        h2, w2 = H // 2, W // 2
        for (ys, xs, dy, dx) in ((0, 0, 8, 12), (0, 1, -8, 12), (1, 0, 8, -12), (1, 1, -8, -12)):
            y0, x0 = ys * h2, xs * w2
            I1[..., y0:y0 + h2, x0:x0 + w2] = base[..., 32 + y0 + dy:32 + y0 + dy + h2, 32 + x0 + dx:32 + x0 + dx + w2]
    u8 = lambda a: (a.clamp(0, 1) * 255).round().to(torch.uint8)
    return torch.stack([u8(I0[0]), u8(I1[0])], 0).to(device)


def frames_from_uint8(u8):
    """[2,3,H,W] uint8 -> [1,3,2,H,W] fp32 in [-1,1]  (run_on_your_images.py:84-87)."""
    return ((u8.float() / 255) * 2 - 1).permute(1, 0, 2, 3).unsqueeze(0).contiguous()


def load_weights(path):
    z = np.load(path)
    return {k: torch.from_numpy(z[k]) for k in z.files}
