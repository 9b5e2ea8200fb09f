"""Helpers with the reference's names (useful.py).  `getmodelconfig` (:163-190) configures the inference path and `MyPWC`
(:104-117) wraps this package's PWCNet; `ScaleIt` (:5-37), `MYPCA` (:41-100) and `distillation_loss` (:119-144) are
TRAINING-TIME helpers (PCA fitting / inspection, the teacher loss): off the hot path, plain torch (the reference fits with a
CuPy SVD), kept so that PCA fitting and inspection scripts written against the reference still run; the reference's drivers
import them by name (main.py:13, utils.py:20, fLDRnet.py:16).  `torch_prints` / `numpy_prints` (:146-161) are the debug printers."""
import numpy as np
import torch
import torch.nn.functional as F

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


class ScaleIt():
    """Per-plane min/max scaler: planes are the last two axes, `free_axes` (2 or 3) leading axes are kept."""

    def __init__(self, name, arr, free_axes):
        assert arr.dim() == free_axes + 2
        self.name, self.arr, self.free_axes = name, arr, free_axes
        self.maxes = arr.amax(dim=(-2, -1), keepdim=True).double()
        self.mins = arr.amin(dim=(-2, -1), keepdim=True).double()

    def scale(self, arr, n=1.0):
        return ((arr - self.mins) / (self.maxes - self.mins)).to(torch.float32)

    def backscale(self, arr):
        return (arr * (self.maxes - self.mins) + self.mins).to(torch.float32)

    def print_mins_maxes(self):
        print("Maxes: ", self.maxes.detach().cpu().numpy())
        print("Mins: ", self.mins.detach().cpu().numpy())


class MYPCA():
    """PCA of row vectors by SVD (training-time fitting of the projection the checkpoint stores as EV8 / Mean8)."""

    def __init__(self, n_components=0):
        self.n_components = n_components
        self.store = dict()

    def fit(self, data):
        self.n_components = self.n_components or data.shape[1]
        self.n = data.shape[0]
        self.mean = data.mean(dim=0)
        _, s, vh = torch.linalg.svd((data - self.mean).cpu(), full_matrices=False)
        self.eigenvectors = vh[:self.n_components]
        self.eigenvalues = s ** 2 / self.n
        self.explained_variance_ratio_ = self.eigenvalues / self.eigenvalues.sum()

    def transform(self, data, device, compsused=0):
        self.mean = torch.as_tensor(self.mean, device=device)
        self.eigenvectors = torch.as_tensor(self.eigenvectors, device=device)
        ev = self.eigenvectors if compsused == 0 else self.eigenvectors[:compsused]
        return (data - self.mean) @ ev.T

    def fit_transform(self, data, device):
        self.fit(data)
        return self.transform(data, device)

    def inverse_transform(self, data):
        return data @ self.eigenvectors + self.mean

    def store_sth(self, toSave, name):
        self.store[name] = toSave


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


def distillation_loss(unref_flow_pyramid, gtflow, device):
    """Teacher-confidence-weighted generalised Charbonnier loss of the coarser flow levels against a teacher flow
    (training only).  Entry 0 of the pyramid (x8 upsampled) only sets the per-pixel confidence p = exp(-0.3 |f - gt|):
    exponent p/2, epsilon 10^(-(10p-1)/3); entries 1.. are resized to (H, H) like the reference (:134-135) and summed."""
    top = F.interpolate(unref_flow_pyramid[0], scale_factor=8, mode='bilinear', align_corners=False).detach()
    side = top.shape[-2]
    loss = torch.tensor(0.0, device=device)
    for sl in (slice(0, 2), slice(2, 4)):
        gt = gtflow[:, sl]
        p = (-0.3 * (top[:, sl] - gt).pow(2).sum(dim=1, keepdim=True).sqrt()).exp()
        alpha, eps = p / 2, 10 ** (-(10 * p - 1) / 3)
        for lvl in unref_flow_pyramid[1:]:
            f = F.interpolate(lvl[:, sl], size=(side, side), mode='bilinear', align_corners=False)
            loss = loss + (((f - gt) ** 2 + eps ** 2) ** alpha).mean()
    return loss


def _prints(arr, name, lib):
    print("------------------------ ARRAY " + name + " PRINTS ------------------------")
    print("shape: ", arr.shape)
    for what in ("min", "max", "mean", "std"):
        print(what + ": ", getattr(lib, what)(arr))


def torch_prints(arr, name=""):
    _prints(arr, name, torch)


def numpy_prints(arr, name=""):
    _prints(arr, name, np)
