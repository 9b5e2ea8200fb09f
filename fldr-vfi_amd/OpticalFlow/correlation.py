"""PWC cost-volume operator with the reference's names (OpticalFlow/correlation.py:415-428), forward and backward,
backed by fldr_correlation_fwd / fldr_correlation_bwd.  Unlike the reference (correlation.py:7-8) the stream is looked up per call."""
import torch

import fldr_hip


class _FunctionCorrelation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second):
        assert first.is_contiguous() and second.is_contiguous()      # correlation.py:302-303
        if not first.is_cuda:
            raise NotImplementedError()                              # correlation.py:343-344
        ctx.save_for_backward(first, second)                         # correlation.py:300
        return fldr_hip.correlation_fwd(first, second)

    @staticmethod
    def backward(ctx, gradOutput):
        first, second = ctx.saved_tensors                            # correlation.py:350-410
        return fldr_hip.correlation_bwd(first, second, gradOutput.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])


def FunctionCorrelation(tensorFirst, tensorSecond):
    return _FunctionCorrelation.apply(tensorFirst, tensorSecond)


class ModuleCorrelation(torch.nn.Module):
    def forward(self, tensorFirst, tensorSecond):
        return _FunctionCorrelation.apply(tensorFirst, tensorSecond)
