"""PWC cost-volume operator with the reference's names (OpticalFlow/correlation.py:415-428), forward only,
backed by fldr_correlation_fwd.  Unlike the reference (correlation.py:7-8) the stream is looked up per call."""
import torch

import fldr_hip


class _FunctionCorrelation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second):
        assert first.is_contiguous() and second.is_contiguous()      # correlation.py:302-303
        if not first.is_cuda:
            raise NotImplementedError()                              # correlation.py:343-344
        return fldr_hip.correlation_fwd(first, second)

    @staticmethod
    def backward(ctx, gradOutput):
        raise NotImplementedError("inference path: correlation backward (correlation.py:114-242) is out of scope")


def FunctionCorrelation(tensorFirst, tensorSecond):
    return _FunctionCorrelation.apply(tensorFirst, tensorSecond)


class ModuleCorrelation(torch.nn.Module):
    def forward(self, tensorFirst, tensorSecond):
        return _FunctionCorrelation.apply(tensorFirst, tensorSecond)
