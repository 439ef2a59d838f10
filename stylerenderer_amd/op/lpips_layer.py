"""One LPIPS layer — normalize_tensor, squared difference, `lin` 1x1 convolution, spatial average — as one fused forward
and one fused backward launch (C ABI sr_lpips_layer_fwd / _bwd, csrc/lpips.hip) instead of ~25 ATen launches.

    lpips_layer(f0 [B, C, H, W], t_normalised [B | 1, C, H, W], lin [1, C, 1, 1] | [C]) -> [B, 1, 1, 1]

Differentiable once, with respect to f0 (the latent-inversion loop optimises the image behind f0; the target and the
learned heads are fixed).  reference lpips/networks_basic.py:62-85, lpips/__init__.py:42-44.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from ._dispatch import on_device_of, require_f32, stream_of

EPS = 1e-10


def supported(f0, t, lin):
    return (f0.device.type == "cuda" and f0.dtype == torch.float32 and f0.dim() == 4 and t.dtype == torch.float32
            and t.shape[1:] == f0.shape[1:] and t.shape[0] in (1, f0.shape[0]) and lin.numel() == f0.shape[1]
            and not t.requires_grad and not lin.requires_grad)


class _LpipsLayer(Function):
    @staticmethod
    def forward(ctx, f0, t, lin):
        require_f32(f0, "lpips_layer")
        f0 = f0.contiguous()
        t = t.contiguous()
        lin = lin.reshape(-1).contiguous()
        b, c, h, w = f0.shape
        hw = h * w
        d = torch.empty(b, dtype=f0.dtype, device=f0.device)
        L = _lib.lib()
        scratch = torch.empty(L.sr_lpips_layer_scratch_floats(b, hw), dtype=f0.dtype, device=f0.device)
        ctx.bstride = c * hw if t.shape[0] == b else 0            # (one target for the whole batch: stride 0)
        with on_device_of(f0):
            rc = L.sr_lpips_layer_fwd(_lib.ptr(d), _lib.ptr(f0), _lib.ptr(t), _lib.ptr(lin), b, c, hw, ctx.bstride,
                                      EPS, _lib.ptr(scratch), stream_of(f0))
        _lib.check(rc, "sr_lpips_layer_fwd")
        ctx.save_for_backward(f0, t, lin)
        return d.view(b, 1, 1, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, gd):
        f0, t, lin = ctx.saved_tensors
        b, c, h, w = f0.shape
        gd = gd.reshape(b).contiguous()
        gf = torch.empty_like(f0)
        with on_device_of(f0):
            rc = _lib.lib().sr_lpips_layer_bwd(_lib.ptr(gf), _lib.ptr(gd), _lib.ptr(f0), _lib.ptr(t), _lib.ptr(lin), b, c,
                                               h * w, ctx.bstride, EPS, stream_of(f0))
        _lib.check(rc, "sr_lpips_layer_bwd")
        return gf, None, None


def lpips_layer(f0, t_normalised, lin):
    return _LpipsLayer.apply(f0, t_normalised, lin)


class _MSE(Function):
    """mean((a - b)^2) -> 0-d, differentiable once w.r.t. a (sr_mse_fwd / _bwd: one launch each way)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty((), dtype=a.dtype, device=a.device)
        with on_device_of(a):
            rc = _lib.lib().sr_mse_fwd(_lib.ptr(out), _lib.ptr(a), _lib.ptr(b), a.numel(), stream_of(a))
        _lib.check(rc, "sr_mse_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        with on_device_of(a):
            rc = _lib.lib().sr_mse_bwd(_lib.ptr(ga), _lib.ptr(g.contiguous()), _lib.ptr(a), _lib.ptr(b), a.numel(), stream_of(a))
        _lib.check(rc, "sr_mse_bwd")
        return ga, None


def mse(a, b):
    """torch.mean((a - b) ** 2) with b fixed; device float32 tensors take the fused kernels."""
    if (a.device.type == "cuda" and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape
            and a.numel() > 0 and not b.requires_grad and a.contiguous().data_ptr() % 16 == 0
            and b.contiguous().data_ptr() % 16 == 0):
        return _MSE.apply(a, b)
    return torch.mean((a - b) ** 2)


class _MaxPool2(Function):
    """F.max_pool2d(x, 2, 2) on even maps — sr_maxpool2_fwd / _bwd (the arg-max is recomputed from the saved input)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        b, c, h, w = x.shape
        out = torch.empty((b, c, h // 2, w // 2), dtype=x.dtype, device=x.device)
        with on_device_of(x):
            rc = _lib.lib().sr_maxpool2_fwd(_lib.ptr(out), _lib.ptr(x), b * c, h, w, stream_of(x))
        _lib.check(rc, "sr_maxpool2_fwd")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        b, c, h, w = x.shape
        gx = torch.empty_like(x)
        with on_device_of(x):
            rc = _lib.lib().sr_maxpool2_bwd(_lib.ptr(gx), _lib.ptr(g.contiguous()), _lib.ptr(x), b * c, h, w, stream_of(x))
        _lib.check(rc, "sr_maxpool2_bwd")
        return gx


def max_pool2(x):
    """2 x 2 / stride 2 max pooling; device float32 tensors with even maps take the library's kernels."""
    if (x.device.type == "cuda" and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] % 2 == 0
            and x.shape[3] % 2 == 0 and x.numel() > 0):
        return _MaxPool2.apply(x)
    return torch.nn.functional.max_pool2d(x, 2, 2)
