"""Inference-side subset of the reference's pca_comp.py: the `DCTParams` carrier the checkpoints pickle
(pca_comp.py:300-305) and `to_pca_diff` (pca_comp.py:473-528) on the gfx950 projection kernel.
`to_pca` (:370-471, PCA fitting with CuPy SVD) and `pca_inverse` (:309-367, reconstruction experiments) are training
code: the names exist so that the reference's drivers (`from pca_comp import DCTParams,to_pca`, main.py:12,
run_on_your_images.py:9; `from pca_comp import pca_inverse,to_pca_diff`, fLDRnet.py:18) import unchanged, and raise
when CALLED — inference never calls them (main.py reaches to_pca only under --phase train)."""
from dataclasses import dataclass

import torch

import fldr_hip


@dataclass
class DCTParams:
    weightMat = torch.zeros((2, 2))
    wiS: int
    components_fraction: float
    data_used: float


def _check(im, params, args, mean_vec):
    if not getattr(args, "mean_vector_norm", True):
        raise NotImplementedError("only the mean_vector_norm=True configuration (useful.py:165) is supported")
    if params.wiS != 8:
        raise NotImplementedError("only 8x8 blocks (wiS=8) are supported")
    k = int(params.wiS * params.wiS * params.components_fraction)
    return k


def to_pca_diff(im, params, args, mean, EV, mean_vec):
    """im [P,H,W] fp32 -> fp64 [P*K,H/8,W/8] in [-1,1] (global min/max), as the reference returns."""
    k = _check(im, params, args, mean_vec)
    _, o64, _ = fldr_hip.pca_project(im, EV.detach()[:k].contiguous(), mean.detach(), mean_vec.detach()[:k].contiguous(),
                                     want_f64=True, want_f32=False)
    return o64


def to_pca_diff_f32(im, params, args, mean, EV, mean_vec, want_spk=False):
    """Same projection, emitting directly the fp32 cast the model applies right after (fLDRnet.py:146): one pass over
    the planes (raw fp64 projections parked in a scratch buffer, rescaled by a streaming kernel).  With want_spk the
    split-packed twin the convolutions consume comes out of the same kernel: -> (fp32, fldr_hip.Spk [1, P*K, h, w])."""
    k = _check(im, params, args, mean_vec)
    o32, _, _, spk = fldr_hip.pca_project_stream(im, EV.detach()[:k].contiguous(), mean.detach(),
                                                 mean_vec.detach()[:k].contiguous(), want_spk=want_spk)
    return (o32, spk) if want_spk else o32


def to_pca_diff_f32_pyramid(ims, params, args, mean, EV, mean_vec, want_spk=False, want_f32=True):
    """to_pca_diff(...).float() for ALL pyramid levels of a forward (the loop of fLDRnet.py:133-146) in two launches:
    ims = [x_l[i].reshape(B*6, h_i, w_i)]; the same EV8 / Mean8 / meanVec8 at every level (fLDRnet.py:135).
    -> list of fp32 [P*K,h,w] (and, with want_spk, the list of split-packed twins; want_f32=False: (None, twins) — the rescale
    launch then writes half the bytes)."""
    k = _check(ims[0], params[0], args, mean_vec)
    for p in params[:len(ims)]:
        if _check(ims[0], p, args, mean_vec) != k:
            raise NotImplementedError("pyramid levels with different numbers of components")
    o32, spk, _ = fldr_hip.pca_project_pyramid(ims, EV.detach()[:k].contiguous(), mean.detach(), mean_vec.detach()[:k].contiguous(),
                                               want_f32=want_f32, want_spk=want_spk)
    return (o32, spk) if want_spk else o32


def to_pca(im, params, components_fraction=0, args=0, pca=0):
    """Training-time PCA fit / transform of 8x8 blocks (pca_comp.py:370-471).  Not part of the inference path."""
    raise NotImplementedError("to_pca fits the block PCA during training; inference projects with to_pca_diff on the "
                              "EV8 / Mean8 / meanVec8 stored in the checkpoint")


def pca_inverse(res, params, pcas, comps_used, cut_back=True, wanted_dim=0, args=0):
    """Reconstruction of image blocks from PCA features (pca_comp.py:309-367), used by training experiments only."""
    raise NotImplementedError("pca_inverse belongs to the training-time reconstruction experiments")
