"""Helpers with the reference's names (useful.py).  `getmodelconfig` (:163-190) configures the inference path and `MyPWC`
(:104-117) wraps this package's PWCNet; `ScaleIt` (:5-37), `MYPCA` (:41-100) and `distillation_loss` (:119-144) are
training-only (out of scope): the names resolve, because the reference's drivers import them (main.py:13,
utils.py:20, fLDRnet.py:16), and raise NotImplementedError when used.  `torch_prints` / `numpy_prints` (:146-161) are the
debug printers."""
import numpy as np
import torch

_PAPER = dict(
    pcanet=True, mean_vector_norm=True, ds_normInput=True, scales=[8, 16, 32, 64], fractions=[4, 16, 64, 256],
    S_trn=3, S_tst=3, dataset="X4K1000FPS", oneEV=True, ref_feat_extrac=True, optimizeEV=True,
    lr_milestones=[70, 120, 170], ExacOneEV=True, takeBestModel=True, allImUp=True, softsplat=True,
    forwendflowloss=True, warp_alpha=0.05, sminterp=True, ownsmooth=True, noResidAddup=True, impmasksoftsplat=True,
    cutoffUnnec=True, fixsmoothtwistup=True, sminterpInpIm=True, patch_size=512, tempbottomflowfix=True,
)


def getmodelconfig(args):
    """Apply the --papermodel settings in place."""
    for k, v in _PAPER.items():
        setattr(args, k, list(v) if isinstance(v, list) else v)
    return args


def _training_only(name, where):
    raise NotImplementedError("useful.%s (%s) is a training-time helper; this package implements the inference path only "
                              "(SURVEY section 2: out of scope)" % (name, where))


class ScaleIt():
    """useful.py:5-37 — per-plane min/max scaler used while fitting / inspecting PCAs (training).  The name resolves so that
    `from useful import *` in the reference's drivers works; constructing it raises."""

    def __init__(self, *args, **kwargs):
        _training_only("ScaleIt", "useful.py:5-37")


class MYPCA():
    """useful.py:41-100 — CuPy SVD fitting of the projection the checkpoint already stores as EV8 / Mean8 (training)."""

    def __init__(self, *args, **kwargs):
        _training_only("MYPCA", "useful.py:41-100")


class MyPWC():
    """PWC-Net teacher wrapper (useful.py:105-117).  Never constructed on the inference path
    (fLDRnet.py:56: mypwc = None); needs external weights that the reference does not ship."""

    def __init__(self, args, cpu=False, checkpoint='./OpticalFlow/pwc-checkpoint.pt'):
        from OpticalFlow.PWCNet import PWCNet
        self.args = args
        self.flow_predictor = PWCNet()
        self.flow_predictor.load_state_dict(torch.load(checkpoint))
        if not cpu:
            self.flow_predictor.to(self.args.gpu)

    def get_flow(self, im0, im1):
        flow = self.flow_predictor(torch.cat([im0, im1], dim=0), torch.cat([im1, im0], dim=0))
        flow01, flow10 = torch.split(flow, im0.shape[0], dim=0)
        return torch.cat([flow10, flow01], dim=1)


def distillation_loss(*args, **kwargs):
    """useful.py:119-144 — teacher-weighted Charbonnier loss against PWC-Net flows (training)."""
    _training_only("distillation_loss", "useful.py:119-144")


def _prints(arr, name, lib):
    print("------------------------ ARRAY " + name + " PRINTS ------------------------")
    print("shape: ", arr.shape)
    for what in ("min", "max", "mean", "std"):
        print(what + ": ", getattr(lib, what)(arr))


def torch_prints(arr, name=""):
    _prints(arr, name, torch)


def numpy_prints(arr, name=""):
    _prints(arr, name, np)
