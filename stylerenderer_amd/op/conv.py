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


def conv_out_size(ih, iw, ksize, stride, pad, transposed):
    if transposed:
        return (ih - 1) * stride + ksize - 2 * pad, (iw - 1) * stride + ksize - 2 * pad
    return (ih + 2 * pad - ksize) // stride + 1, (iw + 2 * pad - ksize) // stride + 1


def conv2d_mfma(x, wt, iscale=None, oscale=None, obias=None, ksize=3, stride=1, pad=1,
                transposed=False):
    require_f32(x, "conv2d_mfma")
    require_f32(wt, "conv2d_mfma weight")
    x = x.contiguous()
    wt = wt.contiguous()
    b, c, ih, iw = x.shape
    taps, cw, n = wt.shape
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
    with on_device_of(x):
        rc = _lib.lib().sr_conv2d_mfma(_lib.ptr(out), _lib.ptr(x), _lib.ptr(wt), _lib.ptr(iscale),
                                       _lib.ptr(oscale), _lib.ptr(obias), b, c, n, ih, iw, oh, ow,
                                       ksize, stride, pad, int(bool(transposed)), stream_of(x))
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
    with on_device_of(x):
        rc = L.sr_conv2d_wgrad_mfma(_lib.ptr(dwt), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(xscale),
                                    _lib.ptr(gscale), b, c, n, ih, iw, oh, ow, ksize, stride, pad,
                                    int(bool(transposed)), _lib.ptr(scratch), stream_of(x))
    _lib.check(rc, "sr_conv2d_wgrad_mfma")
    return dwt
