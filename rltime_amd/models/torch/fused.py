"""Linear + ReLU with the activation in the GEMM epilogue (hipBLASLt) on the GPU.

`relu(x @ W^T + b)` over the IQN head's 1.3 M-row activations is one GEMM plus a
full extra read+write pass for the ReLU in stock PyTorch.  `torch._addmm_activation`
runs bias + ReLU inside the hipBLASLt epilogue; it has no autograd formula, so the
backward is spelled out here (mask by the saved output, two GEMMs, one bias
reduction) — the same math as autograd's linear + relu."""
import torch
import torch.nn.functional as F


class _LinearReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        out = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        x, weight, out = ctx.saved_tensors
        g = torch.ops.aten.threshold_backward(grad.contiguous(), out, 0.0)
        dx = g.mm(weight) if ctx.needs_input_grad[0] else None
        dw = g.t().mm(x) if ctx.needs_input_grad[1] else None
        db = g.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear_relu(x, weight, bias):
    """relu(F.linear(x, weight, bias)) for 2-D x."""
    if x.is_cuda and x.dim() == 2 and x.dtype == weight.dtype and not torch.is_autocast_enabled():
        return _LinearReLU.apply(x, weight, bias)
    return F.relu(F.linear(x, weight, bias))
