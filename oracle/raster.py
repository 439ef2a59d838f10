"""TEST INFRASTRUCTURE — numpy front-end of the C rasterizer oracle (oracle/rasterize_oracle.c).

Mirrors the host-side semantics of the reference's pybind layer and Python wrapper:
  forward_buffers  <- rasterize_forward  (reference op/rasterize.cpp:97-178)
  backward_dcoeff  <- rasterize_backward (reference op/rasterize.cpp:179-241)
  rasterize / rasterize_grads <- Rasterize.forward / .backward (reference op/rasterize.py:17-80)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        sys.path.insert(0, _HERE)
        import build_oracle

        _lib = ctypes.CDLL(build_oracle.build())
        i64, p = ctypes.c_int64, ctypes.c_void_p
        for suf, real in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            fn = getattr(_lib, "oracle_rasterize_forward_" + suf)
            fn.restype = i64
            fn.argtypes = [i64] * 5 + [ctypes.c_int] * 3 + [p] * 5 + [real]
            fn = getattr(_lib, "oracle_rasterize_backward_" + suf)
            fn.restype = i64
            fn.argtypes = [i64] * 4 + [ctypes.c_int] * 2 + [p] * 3 + [real]
        _lib.oracle_rasterize_interp_f32.restype = None
        _lib.oracle_rasterize_interp_f32.argtypes = [i64, i64, p, p, p, p]
    return _lib


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _shapes(v, tri):
    """(b, nv, nf, repeat_v, repeat_f) with the reference's broadcasting rules."""
    repeat_v = v.ndim == 2
    repeat_f = tri.ndim == 2
    if repeat_v:
        b, nv = 1, v.shape[0]
    else:
        b, nv = v.shape[0], v.shape[1]
    if repeat_f:
        nf = tri.shape[0]
    else:
        if not (tri.shape[0] == b or repeat_v):
            raise ValueError("triangles input error")
        b, nf = tri.shape[0], tri.shape[1]
    return b, nv, nf, repeat_v, repeat_f


def forward_buffers(v, tri, h, w=0, perspective=False, eps=1e-6):
    """-> (index int64 [b,h,w,3], coeff [b,h,w,3], zbuf [b,h,w]) ; batch dim dropped when both
    inputs are unbatched, as the reference does."""
    v = np.ascontiguousarray(v)
    tri = np.ascontiguousarray(tri, dtype=np.int64)
    assert v.dtype in (np.float32, np.float64)
    b, nv, nf, rv, rf = _shapes(v, tri)
    h = 1 if h <= 0 else int(h)
    w = h if w <= 0 else int(w)
    eps = abs(eps)
    suf = "f32" if v.dtype == np.float32 else "f64"
    index = np.zeros((b, h, w, 3), np.int64)
    coeff = np.zeros((b, h, w, 3), v.dtype)
    zbuf = np.full((b, h, w), -np.finfo(v.dtype).max, v.dtype)
    getattr(lib(), "oracle_rasterize_forward_" + suf)(
        b, nv, nf, h, w, int(rv), int(rf), int(bool(perspective)),
        _ptr(v), _ptr(tri), _ptr(index), _ptr(coeff), _ptr(zbuf), float(v.dtype.type(eps)))
    if rv and rf:
        return index[0], coeff[0], zbuf[0]
    return index, coeff, zbuf


def backward_dcoeff(v, index, perspective=False, eps=1e-6):
    """-> dcoeff [..., 3, 9]."""
    v = np.ascontiguousarray(v)
    index = np.ascontiguousarray(index, dtype=np.int64)
    rv = v.ndim == 2
    n = v.shape[0] if rv else v.shape[1]
    if index.ndim == 3:
        b, (h, w) = 1, index.shape[:2]
    else:
        b, h, w = index.shape[:3]
    suf = "f32" if v.dtype == np.float32 else "f64"
    d = np.zeros(index.shape + (9,), v.dtype)
    getattr(lib(), "oracle_rasterize_backward_" + suf)(
        b, n, h, w, int(rv), int(bool(perspective)), _ptr(v), _ptr(index), _ptr(d),
        float(v.dtype.type(abs(eps))))
    return d


def rasterize(v, tex, tri, h=256, w=0, perspective=False, eps=1e-6):
    """Interpolated attributes [b,h,w,c] (or [b,h,w] when tex has no channel dim)."""
    index, coeff, _ = forward_buffers(v, tri, h, w, perspective, eps)
    no_ch = tex.ndim == v.ndim - 1
    c = 1 if no_ch else tex.shape[-1]
    flat = np.ascontiguousarray(tex).reshape(-1, c)
    taken = flat[index.reshape(-1)].reshape(index.shape + (c,))
    prod = taken * coeff[..., None]
    out = (prod[..., 0, :] + prod[..., 1, :]) + prod[..., 2, :]
    return out[..., 0] if no_ch else out


def rasterize_grads(v, tex, tri, grad_out, h=256, w=0, perspective=False, eps=1e-6):
    """(grad_v, grad_tex) of `rasterize` for upstream gradient grad_out, accumulated in
    float64 and rounded once (the reference scatters with a float32 sparse matmul whose
    summation order is unspecified, so this side is compared with a tolerance)."""
    index, coeff, _ = forward_buffers(v, tri, h, w, perspective, eps)
    no_ch = tex.ndim == v.ndim - 1
    c = 1 if no_ch else tex.shape[-1]
    flat = np.ascontiguousarray(tex).reshape(-1, c).astype(np.float64)
    g = np.asarray(grad_out, np.float64)
    if no_ch:
        g = g[..., None]
    taken = flat[index.reshape(-1)].reshape(index.shape + (c,))          # [...,3,c]
    dcoeff = backward_dcoeff(v, index, perspective, eps).astype(np.float64)  # [...,3,9]
    dl_dw = (g[..., None, :] * taken).sum(-1)                             # [...,3]
    per_vert = np.einsum("...i,...ij->...j", dl_dw, dcoeff)               # [...,9]
    nrows = flat.shape[0]
    grad_v = np.zeros((nrows, 3), np.float64)
    np.add.at(grad_v, index.reshape(-1), per_vert.reshape(-1, 3))
    grad_t = np.zeros((nrows, c), np.float64)
    np.add.at(grad_t, index.reshape(-1),
              (g[..., None, :] * coeff[..., None].astype(np.float64)).reshape(-1, c))
    return (grad_v.reshape(v.shape).astype(v.dtype),
            grad_t.reshape(tex.shape).astype(tex.dtype))
