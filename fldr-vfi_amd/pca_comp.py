"""Inference-side subset of the reference's pca_comp.py: the `DCTParams` carrier the checkpoints pickle
(pca_comp.py:300-305) and `to_pca_diff` (pca_comp.py:473-528) on the gfx950 projection kernel.
The PCA fitting / reconstruction experiments of the reference are training code and not provided."""
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
