"""Generic-geometry convolution on device tensors — host side of csrc/conv_generic.hip.

The reference's layers take any kernel extent / stride / padding (EqualConv2d: reference layers.py:204-221;
ModulatedConv2d with any kernel_size incl. its up- and down-sampling forms: layers.py:259-323).  The networks of the
benchmarked path only use the geometries of op/conv.py (matrix cores); everything else runs the three direct kernels
here instead of MIOpen.  The three operators are closed under differentiation — each one's two gradients are the other
two — so the Functions below are differentiable to any order (R1 / path-length style double backward included):

    conv(x, w)             d/dx = dgrad(g, w)        d/dw = wgrad(x, g)
    dgrad(g, w)  (= conv_transpose2d as a forward operator, weight [in, out, kh, kw])
                           d/dg = conv(c, w)         d/dw = wgrad(c, g)
    wgrad(x, g)            d/dx = dgrad(g, c)        d/dg = conv(x, c)
"""
import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import on_device_of, stream_of


def _pair(v):
    return (int(v), int(v)) if not isinstance(v, (tuple, list)) else (int(v[0]), int(v[1]))


def supported(x, w):
    return (x.device.type == "cuda" and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 4
            and w.dim() == 4)


def _check(what, *tensors):
    """The kernels walk raw float32 device pointers: anything else (a .double() model, fp16 / bf16 or autocast inputs,
    mixed devices) raises here instead of reading half- or double-sized buffers as float32."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda" or t.dtype != torch.float32:
            raise RuntimeError("%s: float32 device tensors required, got %s on %s" % (what, t.dtype, t.device))
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("%s: tensors on different devices (%s, %s)" % (what, dev, t.device))


def _geom(x_shape, w_shape, stride, padding):
    b, c, ih, iw = x_shape
    n, c2, kh, kw = w_shape
    if c != c2:
        raise RuntimeError("conv2d_generic: input has %d channels, weight expects %d" % (c, c2))
    sy, sx = stride
    py, px = padding
    oh, ow = (ih + 2 * py - kh) // sy + 1, (iw + 2 * px - kw) // sx + 1
    if oh <= 0 or ow <= 0:
        raise RuntimeError("conv2d_generic: kernel %dx%d does not fit a %dx%d input with padding %s" % (kh, kw, ih, iw,
                                                                                                         (py, px)))
    return b, c, n, ih, iw, oh, ow, kh, kw, sy, sx, py, px


def _fwd(x, w, bias, geo):
    b, c, n, ih, iw, oh, ow = geo[:7]
    _check("conv2d_generic", x, w, bias)
    x, w = x.contiguous(), w.contiguous()
    bias = bias.contiguous() if bias is not None else None
    y = torch.empty((b, n, oh, ow), dtype=x.dtype, device=x.device)
    with on_device_of(x):
        rc = _lib.lib().sr_conv2d_generic(_lib.ptr(y), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), *geo, stream_of(x))
    _lib.check(rc, "sr_conv2d_generic")
    return y


def _dgrad(g, w, geo):
    b, c, n, ih, iw = geo[:5]
    _check("conv2d_generic_dgrad", g, w)
    g, w = g.contiguous(), w.contiguous()
    dx = torch.empty((b, c, ih, iw), dtype=g.dtype, device=g.device)
    with on_device_of(g):
        rc = _lib.lib().sr_conv2d_generic_dgrad(_lib.ptr(dx), _lib.ptr(g), _lib.ptr(w), *geo, stream_of(g))
    _lib.check(rc, "sr_conv2d_generic_dgrad")
    return dx


def _wgrad(x, g, geo):
    b, c, n = geo[:3]
    kh, kw = geo[7], geo[8]
    _check("conv2d_generic_wgrad", x, g)
    x, g = x.contiguous(), g.contiguous()
    dw = torch.empty((n, c, kh, kw), dtype=x.dtype, device=x.device)
    with on_device_of(x):
        rc = _lib.lib().sr_conv2d_generic_wgrad(_lib.ptr(dw), _lib.ptr(x), _lib.ptr(g), *geo, stream_of(x))
    _lib.check(rc, "sr_conv2d_generic_wgrad")
    return dw


class _Conv(Function):
    @staticmethod
    def forward(ctx, x, w, bias, geo):
        ctx.geo = geo
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return _fwd(x, w, bias, geo)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        needs = ctx.needs_input_grad
        gx = _DGrad.apply(g, w, ctx.geo) if needs[0] else None
        gw = _WGrad.apply(x, g, ctx.geo) if needs[1] else None
        gb = g.sum((0, 2, 3)) if (ctx.has_bias and needs[2]) else None
        return gx, gw, gb, None


class _DGrad(Function):
    @staticmethod
    def forward(ctx, g, w, geo):
        ctx.geo = geo
        ctx.save_for_backward(g, w)
        return _dgrad(g, w, geo)

    @staticmethod
    def backward(ctx, c):
        g, w = ctx.saved_tensors
        needs = ctx.needs_input_grad
        gg = _Conv.apply(c, w, None, ctx.geo) if needs[0] else None
        gw = _WGrad.apply(c, g, ctx.geo) if needs[1] else None
        return gg, gw, None


class _WGrad(Function):
    @staticmethod
    def forward(ctx, x, g, geo):
        ctx.geo = geo
        ctx.save_for_backward(x, g)
        return _wgrad(x, g, geo)

    @staticmethod
    def backward(ctx, c):
        x, g = ctx.saved_tensors
        needs = ctx.needs_input_grad
        gx = _DGrad.apply(g, c, ctx.geo) if needs[0] else None
        gg = _Conv.apply(x, c, None, ctx.geo) if needs[1] else None
        return gx, gg, None


def conv2d_generic(x, w, bias=None, stride=1, padding=0):
    """F.conv2d(x, w, bias, stride, padding) for float32 device tensors (dilation 1, groups 1) on the HIP kernels."""
    geo = _geom(tuple(x.shape), tuple(w.shape), _pair(stride), _pair(padding))
    return _Conv.apply(x, w, bias, geo)


def conv_transpose2d_generic(x, w, stride=1, padding=0):
    """F.conv_transpose2d(x, w, stride=stride, padding=padding) (weight [in, out, kh, kw], output_padding 0)."""
    sy, sx = _pair(stride)
    py, px = _pair(padding)
    b, cin, h, wd = x.shape
    cin2, cout, kh, kw = w.shape
    if cin != cin2:
        raise RuntimeError("conv_transpose2d_generic: input has %d channels, weight expects %d" % (cin, cin2))
    oh, ow = (h - 1) * sy - 2 * py + kh, (wd - 1) * sx - 2 * px + kw
    # as the data gradient of the convolution [b, cout, oh, ow] -> [b, cin, h, wd]
    geo = (b, cout, cin, oh, ow, h, wd, kh, kw, sy, sx, py, px)
    return _DGrad.apply(x, w, geo)
