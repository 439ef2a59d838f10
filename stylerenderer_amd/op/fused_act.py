"""fused bias + LeakyReLU + gain — host side of the HIP kernels in csrc/fused_bias_act.hip.

Interface parity with the reference (reference op/fused_act.py):
  fused_leaky_relu(input, bias, negative_slope=0.2, scale=2**0.5)      :86-97
  FusedLeakyReLU(channel, negative_slope=0.2, scale=2**0.5) with `.bias` :74-83
  differentiable twice (backward-of-backward through `out`, :20-49).
`fused_bias_act(...)` below is the operator the reference's pybind module exposes
(op/fused_bias_act.cpp:5-33); it calls the C ABI `sr_fused_bias_act` with raw pointers.

Device tensors ALWAYS run the HIP kernels (no fallback: a missing library raises).  CPU tensors
take the plain PyTorch expression the reference itself uses for CPU inputs (op/fused_act.py:87-94);
unlike the reference, which hard-codes 0.2 there, `negative_slope` is honoured.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn import functional as F

from .. import _lib
from ._dispatch import is_device_tensor, mark_inputs, on_device_of, require_f32, stream_of, wanted


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Tensor-level operator (reference op/fused_bias_act.cpp:5-33): empty `bias` / `refer` mean
    "absent".  Device tensors only."""
    if not is_device_tensor(input):
        raise RuntimeError("fused_bias_act: device tensor required (CPU inputs use fused_leaky_relu)")
    require_f32(input, "fused_bias_act")
    x = input.contiguous()
    b = bias.contiguous() if bias is not None else None
    ref = refer.contiguous() if refer is not None else None
    use_bias = int(b is not None and b.numel() > 0)
    use_ref = int(ref is not None and ref.numel() > 0)
    if use_bias:
        require_f32(b, "fused_bias_act bias")
    if use_ref:
        require_f32(ref, "fused_bias_act refer")
        if ref.numel() != x.numel():
            raise RuntimeError("fused_bias_act: refer must have the shape of input")
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    y = torch.empty_like(x)
    with on_device_of(x):
        rc = _lib.lib().sr_fused_bias_act(
            _lib.ptr(y), _lib.ptr(x), _lib.ptr(b) if use_bias else None,
            _lib.ptr(ref) if use_ref else None, int(act), int(grad), float(alpha), float(scale),
            x.numel(), step_b, b.numel() if use_bias else 0, use_bias, use_ref, stream_of(x))
    _lib.check(rc, "sr_fused_bias_act")
    return y


def _act_backward(grad_output, out, negative_slope, scale, want_bias=True):
    """(grad_input, grad_bias) in one sweep (C ABI sr_fused_act_bwd); without the bias gradient (frozen bias: the
    discriminator inside the generator's phase, the LPIPS trunk) one elementwise launch and no reduction."""
    g = grad_output.contiguous()
    o = out.contiguous()
    require_f32(g, "fused_leaky_relu backward")
    n = o.size(0) if o.dim() > 0 else 1
    c = o.size(1) if o.dim() > 1 else 1
    inner = 1
    for i in range(2, o.dim()):
        inner *= o.size(i)
    gx = torch.empty_like(o)
    # an empty input launches nothing: its bias gradient is the empty sum, not uninitialised memory
    gb = (torch.zeros if o.numel() == 0 else torch.empty)(c if want_bias else 0, dtype=o.dtype, device=o.device)
    L = _lib.lib()
    scratch = (torch.empty(L.sr_fused_act_bwd_scratch_floats(n, c, inner), dtype=o.dtype, device=o.device)
               if want_bias else None)
    with on_device_of(o):
        rc = L.sr_fused_act_bwd(_lib.ptr(gx), _lib.ptr(gb) if want_bias else None, _lib.ptr(g), _lib.ptr(o),
                                float(negative_slope), float(scale), n, c, inner,
                                _lib.ptr(scratch), stream_of(o))
    _lib.check(rc, "sr_fused_act_bwd")
    return gx, gb


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale, want_bias=True):
        ctx.save_for_backward(out)
        ctx.set_materialize_grads(False)      # an absent bias cotangent is None (no zero-fill launch), see backward
        ctx.negative_slope = negative_slope
        ctx.scale = scale
        gx, gb = _act_backward(grad_output, out, negative_slope, scale, want_bias)
        if not want_bias:
            ctx.mark_non_differentiable(gb)
        return gx, gb

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        if gradgrad_input is None:
            if gradgrad_bias is None:
                return None, None, None, None, None
            gradgrad_input = torch.zeros_like(out)
        gradgrad_out = fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1,
                                      ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        mark_inputs(ctx, input, bias, negative_slope, scale)
        out = fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope = negative_slope
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        need_bias = wanted(ctx)[1]
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(
            grad_output, out, ctx.negative_slope, ctx.scale, need_bias)
        return grad_input, (grad_bias if need_bias else None), None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    if not is_device_tensor(input):
        shape = [1, bias.shape[0]] + [1] * (input.dim() - bias.dim() - 1)
        return F.leaky_relu(input + bias.view(*shape), negative_slope=negative_slope) * scale
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
