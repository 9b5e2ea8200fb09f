"""PWC-Net with the reference's class name, forward signatures and state-dict keys (OpticalFlow/PWCNet.py:15-325),
its cost volumes computed by the gfx950 correlation kernel (fldr_correlation_fwd).

Status: this network is NOT on the fLDRnet inference path (`DCTXVFInet.mypwc = None`,
fLDRnet.py:56) and its weights (`pwc-checkpoint.pt`) are not shipped with the reference, so its numerics are
"parity unpinned".  It is provided so that `from OpticalFlow.PWCNet import PWCNet` (useful.py:104) resolves and
the correlation operator has its in-network caller; the convolutions here are plain `torch.nn` layers.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import correlation

_LEVEL_CH = [None, None, 81 + 32 + 2 + 2, 81 + 64 + 2 + 2, 81 + 96 + 2 + 2, 81 + 128 + 2 + 2, 81, None]
_DENSE = (128, 128, 96, 64, 32)
_NAMES = ("moduleOne", "moduleTwo", "moduleThr", "moduleFou", "moduleFiv", "moduleSix")


def _lrelu():
    return nn.LeakyReLU(inplace=False, negative_slope=0.1)


class _Extractor(nn.Module):
    """Six stride-2 stages of three 3x3 convs: 3 -> 16, 32, 64, 96, 128, 196 channels (PWCNet.py:20-88)."""

    def __init__(self):
        super().__init__()
        cin = 3
        for name, c in zip(_NAMES, (16, 32, 64, 96, 128, 196)):
            setattr(self, name, nn.Sequential(nn.Conv2d(cin, c, 3, 2, 1), _lrelu(), nn.Conv2d(c, c, 3, 1, 1), _lrelu(),
                                              nn.Conv2d(c, c, 3, 1, 1), _lrelu()))
            cin = c

    def forward(self, x):
        out = []
        for name in _NAMES:
            x = getattr(self, name)(x)
            out.append(x)
        return out


class _Decoder(nn.Module):
    """One pyramid level of the flow decoder (PWCNet.py:93-220)."""

    def __init__(self, level):
        super().__init__()
        prev, cur = _LEVEL_CH[level + 1], _LEVEL_CH[level]
        if level < 6:
            self.moduleUpflow = nn.ConvTranspose2d(2, 2, 4, 2, 1)
            self.moduleUpfeat = nn.ConvTranspose2d(prev + sum(_DENSE), 2, 4, 2, 1)
            self.dblBackward = [None, None, None, 5.0, 2.5, 1.25, 0.625, None][level + 1]
        c = cur
        for name, co in zip(_NAMES[:5], _DENSE):          # DenseNet-style: every block sees all earlier outputs
            setattr(self, name, nn.Sequential(nn.Conv2d(c, co, 3, 1, 1), _lrelu()))
            c += co
        self.moduleSix = nn.Sequential(nn.Conv2d(c, 2, 3, 1, 1))

    @staticmethod
    def backward_warp(x, flow):
        """`Backward` of the reference (PWCNet.py:146-177): normalised-grid bilinear warp with a validity mask."""
        N, _, H, W = flow.shape
        gx = torch.linspace(-1.0, 1.0, W, device=flow.device).view(1, 1, 1, W).expand(N, -1, H, -1)
        gy = torch.linspace(-1.0, 1.0, H, device=flow.device).view(1, 1, H, 1).expand(N, -1, -1, W)
        fl = torch.cat([flow[:, 0:1] / ((x.size(3) - 1.0) / 2.0), flow[:, 1:2] / ((x.size(2) - 1.0) / 2.0)], 1)
        xin = torch.cat([x, flow.new_ones(N, 1, H, W)], 1)
        out = F.grid_sample(xin, (torch.cat([gx, gy], 1) + fl).permute(0, 2, 3, 1), mode='bilinear', padding_mode='zeros',
                            align_corners=False)
        mask = out[:, -1:, :, :]
        mask = torch.where(mask > 0.999, torch.ones_like(mask), torch.zeros_like(mask))
        return out[:, :-1, :, :] * mask

    def forward(self, first, second, previous):
        if previous is None:
            volume = F.leaky_relu(correlation.FunctionCorrelation(first.contiguous(), second.contiguous()), 0.1)
            feat = volume
        else:
            flow = self.moduleUpflow(previous['tensorFlow'])
            upfeat = self.moduleUpfeat(previous['tensorFeat'])
            warped = self.backward_warp(second, flow * self.dblBackward)
            volume = F.leaky_relu(correlation.FunctionCorrelation(first.contiguous(), warped.contiguous()), 0.1)
            feat = torch.cat([volume, first, flow, upfeat], 1)
        for name in _NAMES[:5]:
            feat = torch.cat([getattr(self, name)(feat), feat], 1)
        return {'tensorFlow': self.moduleSix(feat), 'tensorFeat': feat}


class _Refiner(nn.Module):
    """Dilated context network (PWCNet.py:225-249)."""

    def __init__(self):
        super().__init__()
        cin = 81 + 32 + 2 + 2 + sum(_DENSE)
        layers = []
        for co, d in ((128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1)):
            layers += [nn.Conv2d(cin, co, 3, 1, d, d), _lrelu()]
            cin = co
        layers.append(nn.Conv2d(cin, 2, 3, 1, 1, 1))
        self.moduleMain = nn.Sequential(*layers)

    def forward(self, x):
        return self.moduleMain(x)


class PWCNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("_mean", torch.tensor([0.429, 0.431, 0.397]).view(1, 3, 1, 1), persistent=False)
        self.moduleExtractor = _Extractor()
        self.moduleTwo, self.moduleThr, self.moduleFou = _Decoder(2), _Decoder(3), _Decoder(4)
        self.moduleFiv, self.moduleSix = _Decoder(5), _Decoder(6)
        self.moduleRefiner = _Refiner()

    def in_normalize(self, x):
        """transforms.Normalize([0.429, 0.431, 0.397], [1, 1, 1]) (PWCNet.py:18)."""
        return x - self._mean

    def forward(self, tensorFirst, tensorSecond):
        """Frames [B,3,H,W] -> flow [B,2,H,W] in pixels (PWCNet.py:266-301)."""
        a, b = self.in_normalize(tensorFirst), self.in_normalize(tensorSecond)
        H, W = a.size(2), a.size(3)
        Hp, Wp = int(math.floor(math.ceil(H / 64.0) * 64.0)), int(math.floor(math.ceil(W / 64.0) * 64.0))
        a = F.interpolate(a, size=(Hp, Wp), mode='bilinear', align_corners=False)
        b = F.interpolate(b, size=(Hp, Wp), mode='bilinear', align_corners=False)
        flow = 20.0 * F.interpolate(self.forward_pre(a, b), size=(H, W), mode='bilinear', align_corners=False)
        scale = torch.tensor([float(W) / float(Wp), float(H) / float(Hp)], device=flow.device).view(1, 2, 1, 1)
        return flow * scale

    def forward_pre(self, tensorFirst, tensorSecond):
        f, s = self.moduleExtractor(tensorFirst), self.moduleExtractor(tensorSecond)
        est = None
        for dec, k in ((self.moduleSix, -1), (self.moduleFiv, -2), (self.moduleFou, -3), (self.moduleThr, -4),
                       (self.moduleTwo, -5)):
            est = dec(f[k], s[k], est)
        return est['tensorFlow'] + self.moduleRefiner(est['tensorFeat'])
