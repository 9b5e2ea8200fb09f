"""Softmax splatting operator — same module/class/function names and argument meaning as the
reference's softSplat.py (Softsplat: :355-361, FunctionSoftsplat: :320-352, _FunctionSoftsplat: :220-318),
backed by the gfx950 kernels of libfldr_hip.so instead of CuPy-JIT CUDA strings.

Without gradients `FunctionSoftsplat` is one fused call (pre-scale, splat, normalise, post-scale).  When an argument
requires grad it is composed exactly like the reference (:320-352) from torch ops around `_FunctionSoftsplat`, whose
backward runs fldr_softsplat_bwd (the two backward kernels of softSplat.py:54-158 in one pass).
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
        ctx.save_for_backward(input, flow)       # :225
        return fldr_hip.softsplat_fwd(input, flow)

    @staticmethod
    def backward(ctx, gradOutput):
        input, flow = ctx.saved_tensors          # softSplat.py:259-318
        return fldr_hip.softsplat_bwd(input, flow, gradOutput.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])


def FunctionSoftsplat(tenInput, tenFlow, tenMetric, strType):
    assert tenMetric is None or tenMetric.shape[1] == 1
    assert strType in _TYPES
    if not tenInput.is_cuda:
        raise NotImplementedError()
    needs_grad = torch.is_grad_enabled() and (tenInput.requires_grad or tenFlow.requires_grad or
                                              (tenMetric is not None and tenMetric.requires_grad))
    if not needs_grad:
        return fldr_hip.softsplat_fused(tenInput, tenFlow, tenMetric, strType)
    # autograd path: the reference's composition (softSplat.py:324-350) around the differentiable raw splat
    if strType == 'average':
        tenInput = torch.cat([tenInput, tenInput.new_ones(tenInput.shape[0], 1, tenInput.shape[2], tenInput.shape[3])], 1)
    elif strType == 'linear':
        tenInput = torch.cat([tenInput * tenMetric, tenMetric], 1)
    elif strType == 'softmax':
        tenInput = (tenInput + 1) / 2
        if tenMetric is None:
            tenInput = torch.cat([tenInput * 1, tenInput.new_ones(tenInput.shape[0], 1, tenInput.shape[2], tenInput.shape[3])], 1)
        else:
            tenInput = torch.cat([tenInput * tenMetric.exp(), tenMetric.exp()], 1)
    tenOutput = _FunctionSoftsplat.apply(tenInput, tenFlow)
    if strType != 'summation':
        tenNormalize = tenOutput[:, -1:, :, :]
        tenNormalize = torch.where(tenNormalize == 0.0, torch.ones_like(tenNormalize), tenNormalize)   # :346 without the in-place write
        tenOutput = tenOutput[:, :-1, :, :] / tenNormalize
    return (tenOutput - 0.5) * 2


class Softsplat(nn.Module):
    def __init__(self, strType='softmax'):
        super().__init__()
        self.strType = strType

    def forward(self, img, flow, z=None):
        return FunctionSoftsplat(img, flow, z, self.strType)
