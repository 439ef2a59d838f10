"""Skinny matrix products of the style path as three mutually closed autograd Functions over the kernels of
csrc/style_linear.hip — the tensor algebra the RECORDED backward passes of `op.style` are re-derived from
(create_graph=True: path-length regulariser and R1, reference train.py:110-134), so that gradients of any order of
EqualLinear / the demodulation factor stay on this library's kernels instead of rocBLAS:

    mm_nt(a [B,K], w [N,K]) = a @ w^T   [B,N]    sr_linear_fwd   (no bias, no activation)
    mm_nn(g [B,N], w [N,K]) = g @ w     [B,K]    sr_linear_bwd_x
    mm_tn(g [B,N], a [B,K]) = g^T @ a   [N,K]    sr_linear_bwd_w

d(mm_nt) = (mm_nn, mm_tn), d(mm_nn) = (mm_nt, mm_tn), d(mm_tn) = (mm_nt, mm_nn): the set is closed, every derivative
keeps K (the contiguous dimension, a multiple of 4) in place.  Shapes the kernels do not take (K % 4, misaligned rows)
fall to torch.matmul — or raise under SR_STRICT_NATIVE=1 (tests -m gpu, bench.py)."""
import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import mark_inputs, on_device_of, stream_of, strict_native, wanted

_SLOPE, _GAIN = 0.2, 2 ** 0.5


def _rows(t):
    """[rows, cols] fp32 with unit column stride, 16-byte aligned rows (a copy when the view is not)."""
    if t.stride(1) != 1 or t.stride(0) % 4 or t.stride(0) < t.size(1) or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def supported(k, *tensors):
    return (k % 4 == 0 and all(t.device.type == "cuda" and t.dtype == torch.float32 and t.dim() == 2 and
                               0 < t.size(0) <= 65535 for t in tensors))


def _fallback(what, fn, *tensors):
    # (strict mode is about the product's fp32 device path; CPU tensors and the fp64 references of the tests are not it)
    if strict_native() and all(t.device.type == "cuda" and t.dtype == torch.float32 for t in tensors):
        raise RuntimeError("SR_STRICT_NATIVE: %s would run on a library GEMM (shape outside csrc/style_linear.hip)" % what)
    return fn()


class _NT(Function):
    @staticmethod
    def forward(ctx, a, w):
        mark_inputs(ctx, a, w)
        a_, w_ = _rows(a), w.contiguous()
        b, k = a_.shape
        n = w_.size(0)
        y = torch.empty((b, n), dtype=a.dtype, device=a.device)
        with on_device_of(a):
            rc = _lib.lib().sr_linear_fwd(_lib.ptr(y), _lib.ptr(a_), _lib.ptr(w_), None, b, k, n, a_.stride(0), 1.0, 1.0,
                                          0, _SLOPE, _GAIN, stream_of(a))
        _lib.check(rc, "sr_linear_fwd")
        ctx.save_for_backward(a, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        a, w = ctx.saved_tensors
        need_a, need_w = wanted(ctx)
        return (mm_nn(gy, w) if need_a else None), (mm_tn(gy, a) if need_w else None)


class _NN(Function):
    @staticmethod
    def forward(ctx, g, w):
        mark_inputs(ctx, g, w)
        g_, w_ = g.contiguous(), w.contiguous()
        b, n = g_.shape
        k = w_.size(1)
        out = torch.empty((b, k), dtype=g.dtype, device=g.device)
        with on_device_of(g):
            rc = _lib.lib().sr_linear_bwd_x(_lib.ptr(out), _lib.ptr(g_), None, _lib.ptr(w_), b, k, n, 1.0, 0, _SLOPE,
                                            _GAIN, stream_of(g))
        _lib.check(rc, "sr_linear_bwd_x")
        ctx.save_for_backward(g, w)
        return out

    @staticmethod
    def backward(ctx, go):
        g, w = ctx.saved_tensors
        need_g, need_w = wanted(ctx)
        return (mm_nt(go, w) if need_g else None), (mm_tn(g, go) if need_w else None)


class _TN(Function):
    @staticmethod
    def forward(ctx, g, a):
        mark_inputs(ctx, g, a)
        g_, a_ = g.contiguous(), _rows(a)
        b, n = g_.shape
        k = a_.size(1)
        out = torch.empty((n, k), dtype=g.dtype, device=g.device)
        with on_device_of(g):
            rc = _lib.lib().sr_linear_bwd_w(_lib.ptr(out), None, _lib.ptr(g_), None, _lib.ptr(a_), b, k, n, a_.stride(0),
                                            1.0, 1.0, 0, _SLOPE, _GAIN, stream_of(g))
        _lib.check(rc, "sr_linear_bwd_w")
        ctx.save_for_backward(g, a)
        return out

    @staticmethod
    def backward(ctx, go):
        g, a = ctx.saved_tensors
        need_g, need_a = wanted(ctx)
        return (mm_nt(a, go) if need_g else None), (mm_nn(g, go) if need_a else None)


def mm_nt(a, w):
    if supported(a.size(1), a, w):
        return _NT.apply(a, w)
    return _fallback("mm_nt %s x %s" % (tuple(a.shape), tuple(w.shape)), lambda: torch.matmul(a, w.t()), a, w)


def mm_nn(g, w):
    if supported(w.size(1), g, w):
        return _NN.apply(g, w)
    return _fallback("mm_nn %s x %s" % (tuple(g.shape), tuple(w.shape)), lambda: torch.matmul(g, w), g, w)


def mm_tn(g, a):
    if supported(a.size(1), g, a):
        return _TN.apply(g, a)
    return _fallback("mm_tn %s x %s" % (tuple(g.shape), tuple(a.shape)), lambda: torch.matmul(g.t(), a), g, a)
