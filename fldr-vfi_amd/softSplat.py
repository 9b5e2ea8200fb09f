"""Softmax splatting operator — same module/class/function names and argument meaning as the
reference's softSplat.py (Softsplat: :355-361, FunctionSoftsplat: :320-352, _FunctionSoftsplat: :220-318),
backed by the gfx950 kernels of libfldr_hip.so instead of CuPy-JIT CUDA strings.

Inference only: the two backward kernels (softSplat.py:54-158) are training code and out of scope,
so `_FunctionSoftsplat.backward` raises.
"""
import torch
import torch.nn as nn

import fldr_hip

_TYPES = ('summation', 'average', 'linear', 'softmax')


class _FunctionSoftsplat(torch.autograd.Function):
    """Raw summation splat of an already weighted input (softSplat.py:220-259)."""

    @staticmethod
    def forward(ctx, input, flow):
        assert flow.shape[1] == 2
        assert input.shape[2] == flow.shape[2]
        assert input.shape[3] == flow.shape[3]
        if not input.is_cuda:
            raise NotImplementedError()          # softSplat.py:251-252
        return fldr_hip.softsplat_fwd(input, flow)

    @staticmethod
    def backward(ctx, gradOutput):
        raise NotImplementedError("fldr-hip is an inference path: splat backward (softSplat.py:54-158) is out of scope")


def FunctionSoftsplat(tenInput, tenFlow, tenMetric, strType):
    assert tenMetric is None or tenMetric.shape[1] == 1
    assert strType in _TYPES
    if not tenInput.is_cuda:
        raise NotImplementedError()
    return fldr_hip.softsplat_fused(tenInput, tenFlow, tenMetric, strType)


class Softsplat(nn.Module):
    def __init__(self, strType='softmax'):
        super().__init__()
        self.strType = strType

    def forward(self, img, flow, z=None):
        return FunctionSoftsplat(img, flow, z, self.strType)
