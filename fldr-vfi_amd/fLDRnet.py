"""fLDRnet inference model with the reference's class names, constructor/forward signatures, attribute
names and state-dict keys (fLDRnet.py: DCTXVFInet :25-300, DCTVFInet :302-581, PCARefineUNet :584-644),
so that `from fLDRnet import *` in the reference's drivers resolves to this MI355X-native path.

The nn.Conv2d / nn.Parameter objects only CARRY the weights (strict `load_state_dict` of the shipped
checkpoint works); every tensor operation of the forward is a hand-written gfx950 kernel reached through
the C ABI of libfldr_hip.so (fldr_hip.py).  There is no eager/CPU fallback.  Differences in mechanism,
not in results:
  * channel concatenations, nearest upsampling, ReLU, residual adds and `[:, :4]` slices are fused into
    the convolutions (multi-source reads / epilogues);
  * `bwarp` builds no grid / ones tensors on the host (fLDRnet.py:555-561,569 copy ~177 MB H2D per call);
  * the level-0 softmax/blend tail runs in fp64 inside one kernel instead of ~30 fp64 passes;
  * only the test branch exists: `is_training=True` raises (training is out of scope).
"""
import math  # noqa: F401
import numpy as np  # noqa: F401
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401

import fldr_hip
from pca_comp import pca_inverse, to_pca_diff, to_pca_diff_f32, to_pca_diff_f32_pyramid   # noqa: F401  (re-exported like the reference, fLDRnet.py:18)
from softSplat import Softsplat
from useful import torch_prints, numpy_prints, MyPWC  # noqa: F401  (fLDRnet.py:16)

# No __all__: the reference's drivers do `from fLDRnet import *` (main.py:18, run_on_your_images.py:15) and pick up the
# module's public namespace (DCTXVFInet, F, nn, torch, ...) from it; tests/golden/import_contract.json lists what they use.


def _conv3(cin, cout):
    return nn.Conv2d(cin, cout, [3, 3], 1, [1, 1])


def _dparam(*shape):
    p = nn.Parameter(torch.empty(shape, dtype=torch.float64), requires_grad=False)
    return p


class DCTXVFInet(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.device = torch.device('cuda:' + str(args.gpu) if torch.cuda.is_available() else 'cpu')
        self.lrelu = nn.ReLU()
        self.in_channels = args.img_ch
        self.output_size = (args.patch_size, args.patch_size)
        self.output_size_val = (args.validation_patch_size, args.validation_patch_size)
        self.output_size_test = (2160, 4096)
        self.nf = int(args.dctvfi_nf)
        self.base_modules = nn.ModuleList([])
        if args.ref_feat_extrac:
            c = args.dctvfi_nf * args.img_ch * 2
            self.rec_ctx_ds = nn.Sequential(_conv3(args.dctvfi_nf * 6, self.nf * args.img_ch * 2), nn.ReLU(),
                                            _conv3(self.nf * 6, c), nn.ReLU())
            self.base_modules.append(self.rec_ctx_ds)
        self.vfinet = DCTVFInet(args, self.output_size, self.output_size_test, self.output_size_val)
        self.base_modules.append(self.vfinet)
        self.mypwc = None
        if args.optimizeEV:
            sc = [8] * len(args.scales) if args.allImUp else list(args.scales)
            nfe = args.dctvfi_nf
            self.EV8, self.EV16 = _dparam(nfe, sc[0] ** 2), _dparam(nfe, sc[1] ** 2)
            self.EV32 = _dparam(nfe, sc[2] ** 2) if args.S_trn > 1 else None
            self.EV64 = _dparam(nfe, sc[3] ** 2) if args.S_trn > 2 else None
            self.Mean8, self.Mean16 = _dparam(sc[0] ** 2), _dparam(sc[1] ** 2)
            self.Mean32 = _dparam(sc[2] ** 2) if args.S_trn > 1 else None
            self.Mean64 = _dparam(sc[3] ** 2) if args.S_trn > 2 else None
            self.pca_means = [self.Mean8, self.Mean16, self.Mean32, self.Mean64]
            self.EVs = [self.EV8, self.EV16, self.EV32, self.EV64]
            self.mean_vecs = [0 for _ in range(len(args.scales))]
            if args.meanVecParam:
                self.meanVec8, self.meanVec16 = _dparam(nfe), _dparam(nfe)
                self.meanVec32 = _dparam(nfe) if args.S_trn > 1 else None
                self.meanVec64 = _dparam(nfe) if args.S_trn > 2 else None
                self.mean_vecs = [self.meanVec8, self.meanVec16, self.meanVec32, self.meanVec64]
            self.ev_params = [p for p in self.EVs + self.pca_means if p is not None]
        self.used_pcas = None
        self.params = None
        # Pair-invariant cache: PCA features, all flow levels and the splat metrics do not depend on t
        # (fLDRnet.py:396-405, 442-446), yet the reference recomputes them for each of the 7 t values of an 8x
        # interpolation (main.py:833,867).  Opt-in: set `pair_cache = True`; a hit requires the SAME level-0 tensor
        # object at the same version (no content compare: that would cost a device sync and a 283 MB read per miss).
        # Only t-independent RESULTS are cached (flow, z0, z1); the frames themselves are always read from the tensor
        # passed to this call, never from a view saved by an earlier one.
        self.pair_cache = False
        self._pair_state = None

    def _pair_lookup(self, x0):
        st = self._pair_state
        if st is None:
            return None
        ref, ver = st["key"]
        return st if (ref is x0 and ver == x0._version) else None

    # ---- reference API ------------------------------------------------------------------------
    def save_params(self, params):
        if self.params is None:
            self.params = params

    def pick_pca(self, pca):
        raise NotImplementedError("pick_pca installs freshly fitted PCAs during training (fLDRnet.py:225-278); "
                                  "inference reads EV8/Mean8/meanVec8 from the checkpoint")

    def pick_norm_vec(self, pca):
        """fLDRnet.py:279-293: only needed when meanVecParam is False (mean vectors stored outside the state dict)."""
        if self.args.meanVecParam:
            return
        raise NotImplementedError("checkpoints without meanVec parameters are not supported")

    def extract_features(self, pca):
        """rec_ctx_ds(x) + x  (fLDRnet.py:44-49,162): two fused conv kernels."""
        return self._extract_features(pca)[0]

    def _extract_features(self, pca, pca_spk=None):
        """-> (features fp32 NCHW, the same features split-packed or None).  The intermediate activation only exists
        split-packed; the result is written in both layouts by one kernel (fp32 for the splats, packed for conv_flow1)."""
        c0, c2 = self.rec_ctx_ds[0], self.rec_ctx_ds[2]
        if not fldr_hip.use_spk():
            y = fldr_hip.conv2d([pca], c0.weight, c0.bias, relu=True)
            return fldr_hip.conv2d([y], c2.weight, c2.bias, relu=True, residual=pca), None
        y = fldr_hip.conv2d_spk([pca if pca_spk is None else pca_spk], c0.weight, c0.bias, relu=True, want_f32=False, want_spk=True)
        return fldr_hip.conv2d_spk([y], c2.weight, c2.bias, relu=True, residual=pca, want_f32=True, want_spk=True)

    def forward(self, input_gpuList, t_value, normInput=0, is_training=True, validation=False, epoch=0, frameT=None, *, emit_u8=None):
        """input_gpuList: ignored placeholders (the reference overwrites them, fLDRnet.py:134); t_value [B,1];
        normInput: list of S_tst+1 tensors [B,3,2,H/2^i,W/2^i].  Returns (out fp64 [B,3,<=2160,<=4096], flow|None).
        emit_u8 = (H, W) (keyword-only, not in the reference's signature; fldr_harness.interpolate_u8): return the frame cropped to H x W and
        rounded to 8 bits straight from the synthesis kernel where the fused kernel runs (the fp64 frame is then never written); paths that
        cannot honour it return the fp64 frame as ever — the caller looks at the dtype."""
        if is_training:
            raise NotImplementedError("fldr-hip implements the inference (test) branch only")
        # fault flags of EARLIER forwards, read without a synchronisation (two host words the kernels store into): a drop-in caller that
        # never calls fldr_hip.check_range() still gets the exception, one forward late (the reference's CUDA kernels abort on a device-side
        # fault, softSplat.py:25-26); a ring fault additionally turns every later frame into NaN
        fldr_hip.poll_status()
        B2, C2 = t_value.size()
        assert C2 == 1, "t_value shape is [B,]"
        x_l = normInput
        a = self.args
        n_levels = a.S_tst + 1
        i8 = a.scales.index(8)
        if self.params is None:
            raise RuntimeError("call save_params(...) first (main.py:347)")
        t4 = t_value.view(B2, 1, 1, 1)
        state = self._pair_lookup(x_l[0]) if self.pair_cache else None
        if state is None:
            spk = fldr_hip.use_spk()
            B = x_l[0].shape[0]
            nch = a.dctvfi_nf * 6
            # all six projections in two launches (pass A: per-level min / max, pass B: emit): fLDRnet.py:133-146
            # rec_ctx_ds of ALL levels in two launches.  The second convolution adds the PCA features back (fLDRnet.py:162): from their fp32
            # copy (default, fldr_hip.PCA_F32: the parity configuration), or — FLDR_PCA_F32=0, opt-in — from the split-packed tensor (hi + lo:
            # the fp32 feature up to 2^-22 relative, |x| <= 1), in which case the rescale launch writes no fp32 copy (71 MB of 141 per 4K forward).
            levels_batched = bool(a.ref_feat_extrac and spk and B == 1 and fldr_hip.LEVEL_BATCH and fldr_hip.spk_variant() == 1)
            packed_only = levels_batched and not fldr_hip.PCA_F32
            r = to_pca_diff_f32_pyramid([x_l[i].reshape(B * 6, x_l[i].shape[3], x_l[i].shape[4]) for i in range(n_levels)],
                                        self.params, a, self.pca_means[i8], self.EVs[i8], self.mean_vecs[i8], want_spk=spk,
                                        want_f32=not packed_only)
            pcas, pcas_p = r if spk else (r, [None] * n_levels)
            feats = []
            pv, pp = [], []
            for i in range(n_levels):
                h, w = x_l[i].shape[3], x_l[i].shape[4]
                pv.append(pcas[i].view(B, nch, h // 8, w // 8) if pcas is not None else None)
                # [1, 96B, h, w] packed == [B, 96, h, w] packed (12 whole groups per sample)
                pp.append(fldr_hip.Spk(pcas_p[i].buf, (B, nch, h // 8, w // 8)) if spk else None)
            if levels_batched:
                # rec_ctx_ds(x) + x of ALL levels in two launches (the weights are shared, the levels independent: fLDRnet.py:148-162)
                c0, c2 = self.rec_ctx_ds[0], self.rec_ctx_ds[2]
                ys = fldr_hip.conv2d_spk_levels(pp, c0.weight, c0.bias, relu=True, want_f32=False, want_spk=True)
                feats = fldr_hip.conv2d_spk_levels(ys, c2.weight, c2.bias, relu=True, residuals=pp if packed_only else pv, want_f32=True, want_spk=True)
            else:
                for i in range(n_levels):
                    feats.append(self._extract_features(pv[i], pp[i]) if a.ref_feat_extrac else (pv[i], pp[i]))
            flow = None
            for level in range(a.S_tst, -1, -1):                                                       # :210-218
                flow = self.vfinet.estimate_flow(feats[level], flow)
                feats[level] = None
            state = {"key": (x_l[0], x_l[0]._version), "flow0": flow}
            self._pair_state = state if self.pair_cache else None
        out, refined = self.vfinet._synthesise(state["flow0"], x_l[0], t4, validation,
                                               cache=state if self.pair_cache else None, u8_crop=emit_u8)
        return out[:, :, :self.output_size_test[0], :self.output_size_test[1]], refined                # :222


class DCTVFInet(nn.Module):
    def __init__(self, args, output_size, output_size_test, output_size_val):
        super().__init__()
        self.args = args
        self.device = torch.device('cuda:' + str(args.gpu) if torch.cuda.is_available() else 'cpu')
        self.nf = nf = int(args.dctvfi_nf * args.img_ch)
        self.in_channels = 3
        self.output_size, self.output_size_test, self.output_size_val = output_size, output_size_test, output_size_val
        self.softsplat = Softsplat()
        last = 4 if (args.cutoffUnnec and not args.tempbottomflowfix) else 6
        self.conv_flow_bottom = nn.Sequential(_conv3(2 * nf, 2 * nf), nn.ReLU(), _conv3(2 * nf, 2 * nf), nn.ReLU(),
                                              _conv3(2 * nf, 2 * nf), nn.ReLU(), _conv3(2 * nf, nf), nn.ReLU(),
                                              _conv3(nf, last))
        self.conv_flow1 = _conv3(2 * nf, nf)
        self.conv_flow2 = nn.Sequential(_conv3(2 * nf + 4, 2 * nf), nn.ReLU(), _conv3(2 * nf, 2 * nf), nn.ReLU(),
                                        _conv3(2 * nf, nf), nn.ReLU(), _conv3(nf, nf), nn.ReLU(), _conv3(nf, 4))
        self.refine_unet = PCARefineUNet(args)
        self.lrelu = nn.ReLU()
        if args.sminterp:
            self.T_param = nn.Parameter(torch.ones(1, dtype=torch.float64), requires_grad=False)
        if args.impmasksoftsplat:
            self.z_alpha = nn.Parameter(torch.ones(2, dtype=torch.float64))
        self._scalars = None

    def _host_scalars(self):
        """T_param / z_alpha as Python floats, fetched once per parameter version (one D2H sync, not per frame)."""
        key = (self.T_param._version, self.z_alpha._version, self.T_param.data_ptr())
        if self._scalars is None or self._scalars[0] != key:
            za = self.z_alpha.detach().cpu()
            self._scalars = (key, float(self.T_param.detach().cpu()[0]), float(za[0]), float(za[1]))
        return self._scalars[1:]

    @staticmethod
    def _chain(x_srcs, seq, idxs, final_store=None, final_residual=None):
        """conv+ReLU chain over nn.Sequential `seq`; the last index gets no activation.  Activations between the
        convolutions exist only split-packed (fldr_hip.Spk); the chain's result is fp32 NCHW."""
        x = x_srcs
        spk = fldr_hip.use_spk()
        for n, i in enumerate(idxs):
            m = seq[i]
            last = n == len(idxs) - 1
            if spk:
                y = fldr_hip.conv2d_spk(x, m.weight, m.bias, relu=not last, cout_store=final_store if last else None,
                                        residual=final_residual if last else None, want_f32=last, want_spk=not last)
            else:
                y = fldr_hip.conv2d(x, m.weight, m.bias, relu=not last, cout_store=final_store if last else None,
                                    residual=final_residual if last else None)
            x = [y]
        return x[0]

    def forward(self, feat_x, flow_l_prev, t_value, level, is_training, normInput=0, validation=False, epoch=0,
                feat_pyr=[], mypwc=[], orig_images=None, frameT=None):
        if is_training:
            raise NotImplementedError("fldr-hip implements the inference (test) branch only")
        flow_l = self.estimate_flow(feat_x, flow_l_prev)
        if level != 0:
            return flow_l                                                                              # :396-397
        return self._synthesise(flow_l, normInput, t_value, validation)

    def estimate_flow(self, feat_x, flow_l_prev):
        """Flow estimation of one pyramid level (fLDRnet.py:368-391); t-independent."""
        a = self.args
        feat_p = None
        if isinstance(feat_x, tuple):                                  # (fp32 NCHW, split-packed twin) from _extract_features
            feat_x, feat_p = feat_x
        B, C, H, W = feat_x.shape
        half = a.img_ch * a.dctvfi_nf
        feat0, feat1 = feat_x[:, :half], feat_x[:, half:]              # the F4 split of fLDRnet.py:368-370
        spk = fldr_hip.use_spk()
        if spk and feat_p is None:
            feat_p = fldr_hip.spk_pack(feat_x)
        if flow_l_prev is None:
            flow_l = self._chain([feat_p if spk else feat_x], self.conv_flow_bottom, (0, 2, 4, 6, 8), final_store=4)   # :379-380
        else:
            up_p = bw = None
            if spk and fldr_hip.SPLAT_FEATURES != "gather" and H * W > 2304:     # ... and the bounds tables of the two feature splats below come out of the same launch
                up, up_p, bw = fldr_hip.resize_bilinear_spk_bounds(flow_l_prev, H, W, mul=W / flow_l_prev.shape[3])   # :384-385
            elif spk:     # the upsampled flow is consumed as fp32 (splats, residual) and packed (conv_flow2.0): one kernel writes both
                up, up_p = fldr_hip.resize_bilinear_spk(flow_l_prev, H, W, mul=W / flow_l_prev.shape[3])   # :384-385
            else:
                up = fldr_hip.resize_bilinear(flow_l_prev, H, W, mul=W / flow_l_prev.shape[3])         # :384-385
            f1 = self.conv_flow1
            wpair = None
            if spk and C // 2 <= 48 and fldr_hip.SPLAT_FEATURES == "gather":
                # opt-in: both warped feature maps in one deterministic gather launch; they only feed conv_flow1: split-packed
                w1, w0 = fldr_hip.softsplat_gather([feat1, feat0], [up[:, :2], up[:, 2:]], None, "softmax")   # :386-387
            elif spk:
                # both directions in one launch of the fp64-LDS-atomic tile splat: no accumulator, memset or normalisation pass
                # (maps of <= 2304 pixels need no table: every tile walks the whole map)
                pair_batch = B == 1 and half % 8 == 0
                r = fldr_hip.softsplat_acc64([feat1, feat0], [up[:, :2], up[:, 2:]], None, "softmax", want_f32=False,
                                             want_spk=True, spk_batch=pair_batch, bounds_ws=bw)        # :386-387
                if pair_batch:
                    wpair, w1, w0 = r, r.sample(0), r.sample(1)
                else:
                    w1, w0 = r
            else:
                w1 = self.softsplat(feat1, up[:, :2])                                                  # :386
                w0 = self.softsplat(feat0, up[:, 2:])                                                  # :387
            if wpair is not None:
                # conv_flow1(cat(feat0, w1)) and conv_flow1(cat(feat1, w0)) share their weights: ONE launch over a batch of two
                # (feat seen as its two channel halves, the warped maps written side by side above) — twice the units per launch
                pair = fldr_hip.conv2d_spk([feat_p.channel_halves(), wpair], f1.weight, f1.bias, want_f32=False, want_spk=True)
                ca, cb = pair.sample(0), pair.sample(1)
            elif spk:
                ca = fldr_hip.conv2d_spk([feat_p.narrow(0, half), w1], f1.weight, f1.bias, want_f32=False, want_spk=True)
                cb = fldr_hip.conv2d_spk([feat_p.narrow(half, half), w0], f1.weight, f1.bias, want_f32=False, want_spk=True)
            else:
                ca = fldr_hip.conv2d([feat0, w1], f1.weight, f1.bias)
                cb = fldr_hip.conv2d([feat1, w0], f1.weight, f1.bias)
            flow_l = self._chain([ca, cb, up_p if up_p is not None else up], self.conv_flow2, (0, 2, 4, 6, 8), final_residual=up)    # :389-391
        return flow_l

    # ---- level 0 (fLDRnet.py:400-535) ------------------------------------------------------------
    def _synthesise(self, flow_l, x_l, t_value, validation, cache=None, u8_crop=None):
        a = self.args
        B = flow_l.shape[0]
        t4 = t_value.view(B, 1, 1, 1).float()
        T, za0, za1 = self._host_scalars()
        H, W = x_l.shape[3], x_l.shape[4]
        up = x_l.shape[3] / flow_l.shape[2]
        if not float(up).is_integer():
            raise Exception("upscale factor is no integer!!! Upscale factor: " + str(up))            # :412-413
        up = int(up)
        if up == 1:
            raise Exception("Well there should be some upsampling here.")                              # :416-417
        if validation:
            assert (H, W) == tuple(self.output_size_val), "validation crop differs from the input size"
        flow_10_lo, flow_01_lo = flow_l[:, :2], flow_l[:, 2:]
        mask = not a.outMaskLess
        # One kernel (fldr_level0_prep) produces everything between the level-0 flow and the UNet input that is not a
        # splat: the x`up` upsampled flows are never materialised, z0 / z1 (t-independent, cached per pair when enabled)
        # come out of the same pass.  t-scaling happens on the low-resolution flows as in the reference (:404-422).
        inv = cache.get("level0") if cache is not None else None
        I0 = x_l[:, :, 0]              # views of [B,3,2,H,W]: every consumer below takes batch / channel strides, no copies
        I1 = x_l[:, :, 1]
        if inv is not None:
            z0, z1 = inv
        lowres_tables = fldr_hip.SPLAT_BOUNDS == "lowres" and fldr_hip.SPLAT_KERNEL in ("auto", "acc64")
        r = fldr_hip.level0_prep(flow_l, I0, I1, t4, H, W, za0, za1, withmask=mask, want_z=bool(a.impmasksoftsplat) and inv is None)
        if inv is None:
            z0, z1 = r["z0"], r["z1"]                                                                   # :442-446
            if cache is not None:
                cache["level0"] = (z0, z1)
        flow_t0, flow_t1 = r["flow_t0"], r["flow_t1"]                                                   # :404-405,419-422
        if lowres_tables:
            # candidate-source bounds of the two splats from the low-resolution flow their flow_t is the upsampling of; both image
            # splats in one launch (fp64 LDS atomics)
            bw = fldr_hip.splat_bounds_upsampled_pair(flow_l, t4, "images", up, H, W)
            warped0, warped1 = fldr_hip.softsplat_acc64([I0, I1], [flow_t0, flow_t1], [z0, z1] if z0 is not None else None,
                                                        self.softsplat.strType, bounds_ws=bw)          # :449-450
        else:
            warped0 = self.softsplat(I0, flow_t0, z=z0)                                                # :449
            warped1 = self.softsplat(I1, flow_t1, z=z1)                                                # :450
        flowback_0, flowback_1 = r["flowback_0"], r["flowback_1"]                                       # :474-475
        im0_tot, im1_tot = r["im0_tot"], r["im1_tot"]                                                   # :478-479
        srcs = [I0, I1, warped0, warped1, flow_t0, flow_t1, flowback_0, flowback_1, im0_tot, im1_tot]  # :480 (no cat)
        cands = [warped0, warped1, im0_tot, im1_tot, I0, I1]
        unet = self.refine_unet
        if (fldr_hip.DEC23_FUSED and fldr_hip.use_spk() and fldr_hip.CONV_PRECISION == "split" and H % 4 == 0 and W % 4 == 0
                and tuple(unet.dec3.weight.shape) == (6, 16, 3, 3) and tuple(unet.dec2.weight.shape) == (16, 48, 3, 3)):
            # dec2 + dec3 + softmax/T + blend in one persistent kernel: neither dec2's output nor refine_out is ever stored
            dec1p, enc1p = unet.forward_until_dec1(srcs)
            # u8_crop = (H, W) (DCTXVFInet.forward's emit_u8): the cropped frame rounded to 8 bits comes straight out of the kernel's fp64
            # blend instead of the fp64 frame (run_on_your_images.py:100-109 needs nothing else)
            out = fldr_hip.dec23_synth(dec1p, enc1p, unet.dec2.weight, unet.dec2.bias, unet.dec3.weight, unet.dec3.bias, cands, t4, T,
                                       u8_crop=u8_crop)
        elif tuple(unet.dec3.weight.shape) == (6, 16, 3, 3) and H % 2 == 0 and W % 2 == 0:
            # dec3 + softmax/T + blend in one kernel; refine_out (6 full-resolution planes) is never stored
            out = fldr_hip.dec3_synth(unet.forward_until_dec2(srcs, packed_out=fldr_hip.DEC3_MFMA and fldr_hip.use_spk()),
                                      unet.dec3.weight, unet.dec3.bias, cands, t4, T)
        else:
            refine_out = unet(srcs)
            out = fldr_hip.synth_tail(refine_out[:, 0:6], cands, t4, T)                                # :511-524
        flow_out = None
        if a.testgetflowout:
            flow_out = torch.cat([t4 * flow_01_lo, (1 - t4) * flow_10_lo], 1)[:, 0:4]                  # :407,535
        return out, flow_out

    def bwarp(self, x, flo, withmask=True, minus=False):
        """x [B,C,H,W], flo [B,2,H,W] -> backward-warped x (fLDRnet.py:546-581)."""
        return fldr_hip.bwarp(x, flo, withmask)


class PCARefineUNet(nn.Module):
    def __init__(self, args, teach=False):
        super().__init__()
        self.args = args
        self.nf = args.nf
        self.conv1 = _conv3(self.nf, self.nf)      # present in the checkpoint, never called (fLDRnet.py:589-590)
        self.conv2 = _conv3(self.nf, self.nf)
        self.lrelu = nn.ReLU()
        self.NN = nn.UpsamplingNearest2d(scale_factor=2)
        self.input_maps = 26 if args.sminterp else 28
        self.output_maps = 1 + args.img_ch
        if args.sminterp:
            self.output_maps = 3 + 4
        if args.sminterpInpIm:
            self.output_maps += 2
        if args.noResidAddup:
            self.output_maps -= 3
            self.nf = 16
        nf = self.nf
        self.enc1 = nn.Conv2d(self.input_maps, nf, [4, 4], 2, [1, 1])
        self.enc2 = nn.Conv2d(nf, 2 * nf, [4, 4], 2, [1, 1])
        self.enc3 = nn.Conv2d(2 * nf, 4 * nf, [4, 4], 2, [1, 1])
        self.dec0 = _conv3(4 * nf, 4 * nf)
        self.dec1 = _conv3(4 * nf + 2 * nf, 2 * nf)
        self.dec2 = _conv3(2 * nf + nf, nf)
        self.dec3 = _conv3(nf, self.output_maps)

    def _enc3_halves(self):
        """enc3's weights / bias as two halves of 32 output channels (cached copies: the prepacked tables hang on them), or None
        where the packed-source encoder does not take them."""
        w, b = self.enc3.weight, self.enc3.bias
        if tuple(w.shape[:1]) != (64,) or not fldr_hip.use_spk():
            return None
        key = (w._version, w.data_ptr(), b._version, b.data_ptr())
        hit = getattr(self, "_enc3_split", None)
        if hit is None or hit[0] != key:
            parts = [(w.detach()[k:k + 32].contiguous(), b.detach()[k:k + 32].contiguous()) for k in (0, 32)]
            if not all(fldr_hip.s2_spk_ok(p[0]) for p in parts):
                parts = None
            self._enc3_split = hit = (key, parts)
        return hit[1]

    def forward_until_dec1(self, concat):
        """Everything up to and including dec1 + ReLU (fLDRnet.py:621-636) on split-packed activations -> (dec1 packed [B,32,H/4,W/4],
        enc1 packed [B,16,H/2,W/2]): the two inputs of dec2."""
        return self.forward_until_dec2(concat, packed_out=True, stop_before_dec2=True)

    def forward_until_dec2(self, concat, packed_out=False, stop_before_dec2=False):
        """Everything up to and including dec2 + ReLU (fLDRnet.py:621-640), at half resolution."""
        srcs = list(concat) if isinstance(concat, (list, tuple)) else [concat]
        cv = fldr_hip.conv2d
        if fldr_hip.use_spk():
            # the encoders (exact fp32-MFMA stride-2 kernels) emit the split-packed twins the decoder reads; decoder
            # activations only exist split-packed
            cs = fldr_hip.conv2d_spk
            halves = self._enc3_halves() if fldr_hip.ENC3_SPLIT else None
            if fldr_hip.s2_spk_ok(self.enc2.weight):
                # enc2 reads enc1's PACKED output: enc1 writes no fp32 copy of its 16 half-resolution planes (141 MB at 4K)
                enc1p = cv(srcs, self.enc1.weight, self.enc1.bias, stride=2, relu=True, want_f32=False, want_spk=True)
                if halves is not None:
                    enc2, enc2p = None, fldr_hip.conv2d_s2_spk(enc1p, self.enc2.weight, self.enc2.bias, relu=True, want_f32=False, want_spk=True)
                else:
                    enc2, enc2p = fldr_hip.conv2d_s2_spk(enc1p, self.enc2.weight, self.enc2.bias, relu=True, want_f32=True, want_spk=True)
            else:
                halves = None
                enc1, enc1p = cv(srcs, self.enc1.weight, self.enc1.bias, stride=2, relu=True, want_spk=True)
                enc2, enc2p = cv([enc1], self.enc2.weight, self.enc2.bias, stride=2, relu=True, want_spk=True)
            if halves is not None:
                # enc3 (32 -> 64) as two persistent launches of 32 output channels on enc2's PACKED output (64 KB of weights each in
                # LDS; the whole layer's 128 KB only fit the per-tile kernel, which exposes a load round trip per 4-channel chunk):
                # enc2 writes no fp32 copy either; dec0 reads the two halves as two sources
                if fldr_hip.ENC3_PAIR:                       # ... both halves in ONE launch (round 4)
                    out = fldr_hip.conv2d_s2_spk_pair(enc2p, halves, relu=True)
                else:
                    out = [fldr_hip.conv2d_s2_spk(enc2p, w, b, relu=True, want_f32=False, want_spk=True) for (w, b) in halves]
            else:
                out = [cv([enc2], self.enc3.weight, self.enc3.bias, stride=2, relu=True, want_f32=False, want_spk=True)]
            out = cs(out, self.dec0.weight, self.dec0.bias, relu=True, want_f32=False, want_spk=True)
            out = cs([out, enc2p], self.dec1.weight, self.dec1.bias, relu=True, up2=[True, False], want_f32=False, want_spk=True)
            if stop_before_dec2:
                return out, enc1p
            # dec2's output stays split-packed when the fused dec3 + blend kernel consumes it (matrix-core phase convolutions)
            return cs([out, enc1p], self.dec2.weight, self.dec2.bias, relu=True, up2=[True, False],
                      want_f32=not packed_out, want_spk=packed_out)
        enc1 = cv(srcs, self.enc1.weight, self.enc1.bias, stride=2, relu=True)
        enc2 = cv([enc1], self.enc2.weight, self.enc2.bias, stride=2, relu=True)
        out = cv([enc2], self.enc3.weight, self.enc3.bias, stride=2, relu=True)
        out = cv([out], self.dec0.weight, self.dec0.bias, relu=True)
        out = cv([out, enc2], self.dec1.weight, self.dec1.bias, relu=True, up2=[True, False])     # NN + cat (:632-634)
        return cv([out, enc1], self.dec2.weight, self.dec2.bias, relu=True, up2=[True, False])    # :638-640

    def forward(self, concat, feat_dim=0):
        """concat: the 26-channel tensor of fLDRnet.py:480 OR the list of its parts (never materialised)."""
        out = self.forward_until_dec2(concat)
        return fldr_hip.conv2d([out], self.dec3.weight, self.dec3.bias, up2=[True])               # :642-643
