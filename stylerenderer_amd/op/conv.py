"""Dense convolution on the matrix cores — host side of csrc/conv_mfma.hip (C ABI sr_conv2d_mfma).

Low-level operator used by layers.ModulatedConv2d / EqualConv2d for device tensors:

  conv2d_mfma(x, wt, iscale=None, oscale=None, obias=None, ksize=3, stride=1, pad=1, transposed=False)
    x [B,C,IH,IW] fp32, wt [k*k, C, N] (tap-major, N contiguous), iscale [B,C], oscale [B,N], obias [N]
    out[b,n] = oscale[b,n] * sum_{tap,c} wt[tap,c,n] * iscale[b,c] * x[b,c,window(tap)] + obias[n]

It replaces the F.conv2d / F.conv_transpose2d(groups=batch) calls of reference layers.py:301-322.
"""
import torch

from .. import _lib
from ._dispatch import on_device_of, require_f32, stream_of
from . import weight_prep as _wp
from .fused_elem import rowdot


# Optional per-launch timing used by bench.py's roofline leg: when PROFILE is a list, every MFMA
# launch is bracketed by events on the launch stream and (kind, geometry, flops, start, end) appended.
PROFILE = None


def _timed(kind, geom, flops, launch):
    if PROFILE is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = launch()
    e1.record()
    PROFILE.append((kind, geom, flops, e0, e1))
    return rc


def conv_out_size(ih, iw, ksize, stride, pad, transposed):
    if transposed:
        return (ih - 1) * stride + ksize - 2 * pad, (iw - 1) * stride + ksize - 2 * pad
    return (ih + 2 * pad - ksize) // stride + 1, (iw + 2 * pad - ksize) // stride + 1


def conv2d_mfma(x, wt, iscale=None, oscale=None, obias=None, ksize=3, stride=1, pad=1,
                transposed=False):
    require_f32(x, "conv2d_mfma")
    require_f32(wt, "conv2d_mfma weight")
    x = x.contiguous()
    b, c, ih, iw = x.shape
    taps, cw, n = wt.shape
    # the kernel reads weight rows as 16-byte vectors: row pitch a multiple of 4 floats (views of the
    # padded buffers of op.weight_prep pass through untouched)
    if (wt.stride(2) == 1 and wt.stride(1) % 4 == 0 and wt.stride(1) >= n and wt.stride(0) == cw * wt.stride(1)
            and wt.data_ptr() % 16 == 0):
        ldw = wt.stride(1)
    else:
        ldw = (n + 3) // 4 * 4
        wt = torch.nn.functional.pad(wt, (0, ldw - n)) if ldw != n else wt.contiguous()
    if taps != ksize * ksize or cw != c:
        raise RuntimeError("conv2d_mfma: weight must be [k*k, C, N]; got %s for C=%d k=%d"
                           % (tuple(wt.shape), c, ksize))
    oh, ow = conv_out_size(ih, iw, ksize, stride, pad, transposed)
    for t, shape, name in ((iscale, (b, c), "iscale"), (oscale, (b, n), "oscale"), (obias, (n,), "obias")):
        if t is not None:
            require_f32(t, "conv2d_mfma " + name)
            if tuple(t.shape) != shape or not t.is_contiguous():
                raise RuntimeError("conv2d_mfma: %s must be contiguous %s" % (name, shape))
    out = torch.empty((b, n, oh, ow), dtype=x.dtype, device=x.device)
    gh, gw = (ih, iw) if transposed else (oh, ow)
    flops = 2.0 * b * gh * gw * c * n * ksize * ksize
    L = _lib.lib()
    nscr = L.sr_conv2d_scratch_floats(b, c, n, ih, iw, oh, ow, ksize, stride, pad, int(bool(transposed)))
    scratch = torch.empty(nscr, dtype=x.dtype, device=x.device) if nscr > 0 else None   # split-K (small maps)
    with on_device_of(x):
        rc = _timed("conv", (ksize, stride, int(bool(transposed)), b, c, n, gh, gw), flops,
                    lambda: L.sr_conv2d_mfma(
                        _lib.ptr(out), _lib.ptr(x), _lib.ptr(wt), _lib.ptr(iscale), _lib.ptr(oscale),
                        _lib.ptr(obias), b, c, n, ldw, ih, iw, oh, ow, ksize, stride, pad,
                        int(bool(transposed)), _lib.ptr(scratch), stream_of(x)))
    _lib.check(rc, "sr_conv2d_mfma")
    return out


def conv2d_wgrad_mfma(x, gy, xscale=None, gscale=None, ksize=3, stride=1, pad=1, transposed=False):
    """dwt [k*k, C, N] = sum over batch and pixels of (xscale*x)[window] * (gscale*gy)."""
    require_f32(x, "conv2d_wgrad_mfma")
    require_f32(gy, "conv2d_wgrad_mfma grad")
    x = x.contiguous()
    gy = gy.contiguous()
    b, c, ih, iw = x.shape
    b2, n, oh, ow = gy.shape
    if b2 != b or (oh, ow) != conv_out_size(ih, iw, ksize, stride, pad, transposed):
        raise RuntimeError("conv2d_wgrad_mfma: shape mismatch x %s gy %s" % (tuple(x.shape), tuple(gy.shape)))
    for t, shape, name in ((xscale, (b, c), "xscale"), (gscale, (b, n), "gscale")):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != torch.float32):
            raise RuntimeError("conv2d_wgrad_mfma: %s must be contiguous float32 %s" % (name, shape))
    L = _lib.lib()
    nfl = L.sr_conv2d_wgrad_scratch_floats(b, c, n, ih, iw, oh, ow, ksize, stride, pad, int(bool(transposed)))
    if nfl < 0:
        raise RuntimeError("conv2d_wgrad_mfma: unsupported geometry")
    scratch = torch.empty(nfl, dtype=torch.float32, device=x.device)
    dwt = torch.empty((ksize * ksize, c, n), dtype=torch.float32, device=x.device)
    gh, gw = (ih, iw) if transposed else (oh, ow)
    flops = 2.0 * b * gh * gw * c * n * ksize * ksize
    with on_device_of(x):
        rc = _timed("wgrad", (ksize, stride, int(bool(transposed)), b, c, n, gh, gw), flops,
                    lambda: L.sr_conv2d_wgrad_mfma(
                        _lib.ptr(dwt), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(xscale), _lib.ptr(gscale),
                        b, c, n, ih, iw, oh, ow, ksize, stride, pad, int(bool(transposed)),
                        _lib.ptr(scratch), stream_of(x)))
    _lib.check(rc, "sr_conv2d_wgrad_mfma")
    return dwt


# ------------------------------------------------------------------------------------------------
# Differentiable operators.  Two mutually recursive autograd Functions close the algebra:
#   ConvFn   y  = oscale * conv(iscale * x, wt) + bias
#   WgradFn  dW = sum_b corr(iscale * x, oscale * g)
# d(ConvFn)/dx is a ConvFn with the adjoint geometry, d(ConvFn)/dwt is a WgradFn, and both
# derivatives of WgradFn are ConvFn's — so gradients of any order (path-length regulariser,
# R1: reference train.py:110-134) run on the same two MFMA kernels.
_ADJOINT = {"c3": "c3", "c3s2": "t3s2", "t3s2": "c3s2", "c1": "c1"}
_GEOM = {  # name -> (ksize, stride, pad, transposed)
    "c3": (3, 1, 1, False), "c3s2": (3, 2, 0, False), "t3s2": (3, 2, 0, True), "c1": (1, 1, 0, False),
    "c1s2": (1, 2, 0, False),
}


def adjoint_weight(wt, geom):
    """Weights of the data-gradient convolution: channels swapped; taps reversed for the
    stride-1 3x3 correlation (the strided pair c3s2 <-> t3s2 keeps tap order)."""
    return _wp.adjoint(wt, geom == "c3")


def _bc(s):
    return s[:, :, None, None]


class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wt, iscale, oscale, bias, geom):
        k, stride, pad, tr = _GEOM[geom]
        out = conv2d_mfma(x, wt, iscale, oscale, bias, k, stride, pad, tr)
        ctx.geom = geom
        ctx.has = (iscale is not None, oscale is not None, bias is not None)
        ctx.save_for_backward(x, wt, iscale, oscale, bias, out if oscale is not None else None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wt, iscale, oscale, bias, out = ctx.saved_tensors
        need_x, need_w, need_is, need_os, need_b = ctx.needs_input_grad[:5]
        geom = ctx.geom
        g = g.contiguous()
        gx = gw = gis = gos = gb = None
        if need_x or need_is:
            if geom == "c1s2":
                inner = ConvFn.apply(g, adjoint_weight(wt, "c1"), oscale, None, None, "c1")
                dxu = torch.zeros_like(x)
                dxu[:, :, ::2, ::2] = inner
            else:
                dxu = ConvFn.apply(g, adjoint_weight(wt, geom), oscale, None, None, _ADJOINT[geom])
            # style gradient sum_p x*dxu and dx = s*dxu in one sweep (csrc/fused_elem.hip)
            if need_is and need_x and iscale is not None:
                gis, gx = rowdot(x, dxu, iscale)
            else:
                if need_is:
                    gis = rowdot(x, dxu)
                if need_x:
                    gx = dxu * _bc(iscale) if iscale is not None else dxu
        if need_w:
            gw = WgradFn.apply(x, g, iscale, oscale, geom)
        if need_os:
            y0 = out - bias[None, :, None, None] if bias is not None else out
            gos = rowdot(g, y0) / oscale
        if need_b:
            gb = g.sum((0, 2, 3))
        return gx, gw, gis, gos, gb, None


class WgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, iscale, oscale, geom):
        k, stride, pad, tr = _GEOM[geom]
        ctx.geom = geom
        ctx.save_for_backward(x, g, iscale, oscale)
        return conv2d_wgrad_mfma(x, g, iscale, oscale, k, stride, pad, tr)

    @staticmethod
    def backward(ctx, gg):
        """gg = d/d(dW) [k*k, C, N]."""
        x, g, iscale, oscale = ctx.saved_tensors
        need_x, need_g, need_is, need_os = ctx.needs_input_grad[:4]
        geom = ctx.geom
        gg = gg.contiguous()
        gx = g_g = gis = gos = None
        if need_x or need_is:
            if geom == "c1s2":
                inner = ConvFn.apply(g, adjoint_weight(gg, "c1"), oscale, None, None, "c1")
                dxu = torch.zeros_like(x)
                dxu[:, :, ::2, ::2] = inner
            else:
                dxu = ConvFn.apply(g, adjoint_weight(gg, geom), oscale, None, None, _ADJOINT[geom])
            if need_is:
                gis = rowdot(x, dxu)
            if need_x:
                gx = dxu * _bc(iscale) if iscale is not None else dxu
        if need_g or need_os:
            dgu = ConvFn.apply(x, gg, iscale, None, None, geom)
            if need_os:
                gos = rowdot(g, dgu)
            if need_g:
                g_g = dgu * _bc(oscale) if oscale is not None else dgu
        return gx, g_g, gis, gos, None


def conv2d(x, wt, iscale=None, oscale=None, bias=None, geom="c3"):
    """Differentiable (to any order) MFMA convolution; see ConvFn."""
    return ConvFn.apply(x, wt, iscale, oscale, bias, geom)
