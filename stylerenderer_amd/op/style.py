"""Skinny linear algebra of the style path — host side of csrc/style_linear.hip.

    equal_linear(x [B,K], weight [N,K], bias [N]|None, wscale, bscale, act)   EqualLinear (+ fused lrelu)
    demod_scale(s [B,Ci], wsq [Ci,Co], eps)                                   rsqrt(s^2 @ wsq + eps)

First-order gradients run on the fused kernels (two launches each).  When the backward pass is
itself being recorded (create_graph=True: the path-length regulariser and R1, reference
train.py:110-134) the vector-Jacobian product is re-derived from the defining tensor algebra under
autograd, which is differentiable to any order.
"""
import torch
from torch.autograd import Function
from torch.nn import functional as F

from .. import _lib
from . import smallmm as _mm
from ._dispatch import mark_inputs, on_device_of, stream_of, wanted

LRELU_SLOPE = 0.2
LRELU_GAIN = 2 ** 0.5


def linear_supported(x, weight):
    return (x.device.type == "cuda" and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.dim() == 2 and weight.dim() == 2 and x.size(1) % 4 == 0 and 0 < x.size(0) <= 65535
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.stride(0) >= x.size(1)
            and x.data_ptr() % 16 == 0)


def _linear_composite(x, weight, bias, wscale, bscale, act):
    """The defining algebra of `_Linear` for its RECORDED backward; the product runs on op.smallmm (this library's
    kernels, closed under differentiation), not on F.linear / rocBLAS."""
    t = _mm.mm_nt(x, weight) * wscale
    if bias is not None:
        t = t + bias * bscale
    return F.leaky_relu(t, LRELU_SLOPE) * LRELU_GAIN if act else t


def _vjp(outputs, inputs, grads, needs):
    """grad of `outputs` w.r.t. the `inputs` flagged in `needs`, differentiable."""
    sel = [t for t, n in zip(inputs, needs) if n and t is not None]
    got = iter(torch.autograd.grad(outputs, sel, grads, create_graph=True, allow_unused=True)) if sel else iter(())
    return [next(got) if (n and t is not None) else None for t, n in zip(inputs, needs)]


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, wscale, bscale, act):
        mark_inputs(ctx, x, weight, bias, wscale, bscale, act)
        w = weight.contiguous()
        b, k = x.shape
        n = w.size(0)
        y = torch.empty((b, n), dtype=x.dtype, device=x.device)
        with on_device_of(x):
            rc = _lib.lib().sr_linear_fwd(_lib.ptr(y), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), b, k, n,
                                          x.stride(0), wscale, bscale, int(act), LRELU_SLOPE, LRELU_GAIN,
                                          stream_of(x))
        _lib.check(rc, "sr_linear_fwd")
        ctx.save_for_backward(x, weight, bias, y)
        ctx.cfg = (float(wscale), float(bscale), bool(act))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, y = ctx.saved_tensors
        wscale, bscale, act = ctx.cfg
        needs = wanted(ctx)[:3]
        if torch.is_grad_enabled():
            with torch.enable_grad():
                # aliases: gradients stop at the node boundary even if the inputs share history outside it
                xa, wa, ba = [t.view_as(t) if t is not None else None for t in (x, weight, bias)]
                out = _linear_composite(xa, wa, ba, wscale, bscale, act)
                gx, gw, gb = _vjp(out, (xa, wa, ba), gy, needs)
            return gx, gw, gb, None, None, None
        gy = gy.contiguous()
        w = weight.contiguous()
        b, k = x.shape
        n = w.size(0)
        L = _lib.lib()
        gx = gw = gb = None
        with on_device_of(x):
            if needs[0]:
                gx = torch.empty((b, k), dtype=x.dtype, device=x.device)
                _lib.check(L.sr_linear_bwd_x(_lib.ptr(gx), _lib.ptr(gy), _lib.ptr(y), _lib.ptr(w), b, k, n, wscale,
                                             int(act), LRELU_SLOPE, LRELU_GAIN, stream_of(x)), "sr_linear_bwd_x")
            if needs[1] or (bias is not None and needs[2]):
                gw = torch.empty_like(w)
                gb = torch.empty_like(bias) if bias is not None and needs[2] else None
                _lib.check(L.sr_linear_bwd_w(_lib.ptr(gw), _lib.ptr(gb), _lib.ptr(gy), _lib.ptr(y), _lib.ptr(x), b,
                                             k, n, x.stride(0), wscale, bscale, int(act), LRELU_SLOPE, LRELU_GAIN,
                                             stream_of(x)), "sr_linear_bwd_w")
                if not needs[1]:
                    gw = None
        return gx, gw, gb, None, None, None


def equal_linear(x, weight, bias, wscale, bscale, act=False):
    """act(wscale * x @ weight^T + bscale * bias); act = leaky_relu(0.2) * sqrt(2) (reference
    layers.py:236-239 with op/fused_act.py:86-97) or identity."""
    return _Linear.apply(x, weight, bias, float(wscale), float(bscale), bool(act))


def demod_supported(s, wsq):
    return (s.device.type == "cuda" and s.dtype == torch.float32 and wsq.dtype == torch.float32 and s.dim() == 2
            and wsq.dim() == 2 and wsq.size(1) % 4 == 0 and 0 < s.size(0) <= 65535 and s.size(1) == wsq.size(0))


def _demod_composite(s, wsq, eps):
    """The defining algebra of `_Demod` for its RECORDED backward (op.smallmm instead of torch.matmul)."""
    return torch.rsqrt(_mm.mm_nn(s * s, wsq) + eps)


class _Demod(Function):
    @staticmethod
    def forward(ctx, s, wsq, eps):
        mark_inputs(ctx, s, wsq, eps)
        s_, w_ = s.contiguous(), wsq.contiguous()
        b, ci = s_.shape
        co = w_.size(1)
        d = torch.empty((b, co), dtype=s.dtype, device=s.device)
        with on_device_of(s):
            rc = _lib.lib().sr_demod_fwd(_lib.ptr(d), _lib.ptr(s_), _lib.ptr(w_), b, ci, co, eps, stream_of(s))
        _lib.check(rc, "sr_demod_fwd")
        ctx.save_for_backward(s, wsq, d)
        ctx.eps = float(eps)
        return d

    @staticmethod
    def backward(ctx, gd):
        s, wsq, d = ctx.saved_tensors
        needs = wanted(ctx)[:2]
        if torch.is_grad_enabled():
            with torch.enable_grad():
                sa, wa = s.view_as(s), wsq.view_as(wsq)
                out = _demod_composite(sa, wa, ctx.eps)
                gs, gw = _vjp(out, (sa, wa), gd, needs)
            return gs, gw, None
        gd = gd.contiguous()
        s_, w_ = s.contiguous(), wsq.contiguous()
        b, ci = s_.shape
        co = w_.size(1)
        gs = torch.empty_like(s_) if needs[0] else None
        gw = torch.empty_like(w_) if needs[1] else None
        if gs is not None or gw is not None:
            with on_device_of(s):
                rc = _lib.lib().sr_demod_bwd(_lib.ptr(gs), _lib.ptr(gw), _lib.ptr(gd), None, _lib.ptr(s_),
                                             _lib.ptr(d), _lib.ptr(w_), b, ci, co, stream_of(s))
            _lib.check(rc, "sr_demod_bwd")
        return gs, gw, None


def demod_scale(s, wsq, eps):
    """rsqrt(s^2 @ wsq + eps): the demodulation factor of reference layers.py:298-300 for shared weights."""
    return _Demod.apply(s, wsq, float(eps))
