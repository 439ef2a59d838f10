"""3DMM triangle rasterizer — host side of csrc/rasterize.hip.

Interface parity with the reference:
  rasterize(v, tex, tri, h=256, w=0, perspective=False, eps=1e-6) -> [b,h,w,c] (or [b,h,w])
                                                    reference op/rasterize.py:17-82
  rasterize_op.forward(v, tri, h, w, perspective, eps) -> [index int64, coeff]
  rasterize_op.backward(v, index, perspective, eps)    -> dcoeff [..., 3, 9]
                                                    reference op/rasterize.cpp:97-245 (pybind layer)
Shape / dtype validation follows the pybind layer (batched or shared vertices and topology,
h<=0 -> 1, w<=0 -> h, float32/float64 + int64 only); violations raise RuntimeError like AT_ASSERTM.

Device tensors run the deterministic HIP rasterizer, whose outputs equal the reference's
sequential CPU loops bit for bit.  The autograd Function fuses what the reference does in Python:
attribute interpolation happens in the resolve pass, and the backward scatters straight into
grad_v / grad_tex instead of building a COO matrix per call (reference op/rasterize.py:46-77).
There is no CPU implementation in the product: CPU tensors raise (the CPU restatement lives in
oracle/ and is test infrastructure).
"""
import types

import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import is_device_tensor, on_device_of, stream_of


def _geometry(vertices, triangles):
    """(b, nv, nf, repeat_v, repeat_f) with the reference's rules (op/rasterize.cpp:103-121)."""
    if vertices.dim() == 2 and vertices.size(-1) == 3:
        repeat_v, b, nv = True, 1, vertices.size(0)
    elif vertices.dim() == 3 and vertices.size(-1) == 3:
        repeat_v, b, nv = False, vertices.size(0), vertices.size(1)
    else:
        raise RuntimeError("vertices input error")
    if triangles.dim() == 3 and triangles.size(2) == 3 and (triangles.size(0) == b or repeat_v):
        repeat_f, b, nf = False, triangles.size(0), triangles.size(1)
    elif triangles.dim() == 2 and triangles.size(1) == 3:
        repeat_f, nf = True, triangles.size(0)
    else:
        raise RuntimeError("triangles input error")
    if not vertices.is_contiguous():
        raise RuntimeError("vertices input error")
    if not triangles.is_contiguous():
        raise RuntimeError("triangles input error")
    return b, nv, nf, repeat_v, repeat_f


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise RuntimeError(" type error")


def _forward_impl(vertices, triangles, height, width, perspective, eps, tex=None, want_z=False):
    if not (is_device_tensor(vertices) and is_device_tensor(triangles)):
        if vertices.device.type != triangles.device.type:
            raise RuntimeError(" cuda input error")
        raise RuntimeError("rasterize: device tensors required (the CPU path is oracle/raster.py)")
    if triangles.dtype != torch.int64:
        raise RuntimeError(" type error")
    suf = _suffix(vertices)
    b, nv, nf, rv, rf = _geometry(vertices, triangles)
    h = 1 if height <= 0 else int(height)
    w = h if width <= 0 else int(width)
    dev, dt = vertices.device, vertices.dtype
    index = torch.empty((b, h, w, 3), dtype=torch.int64, device=dev)
    coeff = torch.empty((b, h, w, 3), dtype=dt, device=dev)
    zbuf = torch.empty((b, h, w), dtype=dt, device=dev) if want_z else None
    attr, tex_c, tex_flat = None, 0, None
    if tex is not None:
        tex_c = 1 if tex.dim() == vertices.dim() - 1 else int(tex.shape[-1])
        tex_flat = tex.contiguous().view(-1, tex_c)
        if tex_flat.dtype != dt:
            raise RuntimeError(" type error")
        attr = torch.empty((b, h, w, tex_c), dtype=dt, device=dev)
    L = _lib.lib()
    work = torch.empty(L.sr_rasterize_scratch_bytes(b, h, w, int(suf == "f64")), dtype=torch.uint8,
                       device=dev)
    with on_device_of(vertices):
        rc = getattr(L, "sr_rasterize_forward_" + suf)(
            b, nv, nf, h, w, int(rv), int(rf), int(bool(perspective)), _lib.ptr(vertices),
            _lib.ptr(triangles), _lib.ptr(index), _lib.ptr(coeff), _lib.ptr(zbuf), abs(float(eps)),
            _lib.ptr(tex_flat), tex_c, _lib.ptr(attr), _lib.ptr(work), stream_of(vertices))
    _lib.check(rc, "sr_rasterize_forward")
    if rv and rf:
        index, coeff = index[0], coeff[0]
        zbuf = zbuf[0] if zbuf is not None else None
        attr = attr[0] if attr is not None else None
    return index, coeff, zbuf, attr


def forward(vertices, triangles, height, width, perspective=False, eps=1e-9):
    index, coeff, _, _ = _forward_impl(vertices, triangles, height, width, perspective, eps)
    return [index, coeff]


def forward_with_depth(vertices, triangles, height, width, perspective=False, eps=1e-9):
    """Extension used by tests: also returns the z-buffer the reference keeps internal."""
    index, coeff, zbuf, _ = _forward_impl(vertices, triangles, height, width, perspective, eps,
                                          want_z=True)
    return index, coeff, zbuf


def backward(vertices, index, perspective=False, eps=1e-9):
    if not (is_device_tensor(vertices) and is_device_tensor(index)):
        raise RuntimeError(" cuda error")
    if index.dtype != torch.int64:
        raise RuntimeError(" type error")
    suf = _suffix(vertices)
    rv = vertices.dim() == 2
    n = vertices.size(0) if rv else vertices.size(1)
    if index.dim() == 3 and rv:
        b, (h, w) = 1, index.shape[:2]
    elif index.dim() == 4:
        b, h, w = index.shape[:3]
    else:
        raise RuntimeError("index input error")
    if index.size(-1) != 3 or not index.is_contiguous() or not vertices.is_contiguous():
        raise RuntimeError("index input error")
    dcoeff = torch.empty(tuple(index.shape) + (9,), dtype=vertices.dtype, device=vertices.device)
    with on_device_of(vertices):
        rc = getattr(_lib.lib(), "sr_rasterize_backward_" + suf)(
            b, n, h, w, int(rv), int(bool(perspective)), _lib.ptr(vertices), _lib.ptr(index),
            _lib.ptr(dcoeff), abs(float(eps)), stream_of(vertices))
    _lib.check(rc, "sr_rasterize_backward")
    return dcoeff


# the reference's extension module object, by name
rasterize_op = types.SimpleNamespace(forward=forward, backward=backward)


class Rasterize(Function):
    @staticmethod
    def forward(ctx, v, tex, tri, h, w, perspective, eps):
        v = v.contiguous()
        tri = tri.contiguous()
        ind, coeff, _, out = _forward_impl(v, tri, h, w, perspective, eps, tex=tex)
        ctx.save_for_backward(v, tex, ind, coeff)
        ctx.perspective = perspective
        ctx.eps = eps
        ctx.no_channel = tex.dim() == v.dim() - 1
        return out[..., 0] if ctx.no_channel else out

    @staticmethod
    def backward(ctx, grad_out):
        v, tex, ind, coeff = ctx.saved_tensors
        need_v, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_v or need_t):
            return (None,) * 7
        if v.dim() != 3:
            raise RuntimeError("rasterize backward: batched vertices [b, n, 3] required")
        suf = _suffix(v)
        b, nv = v.size(0), v.size(1)
        h, w = ind.shape[-3], ind.shape[-2]
        c = 1 if ctx.no_channel else int(tex.shape[-1])
        go = grad_out.contiguous()
        tex_c = tex.contiguous()
        grad_v = torch.zeros_like(v) if need_v else None
        grad_t = torch.zeros_like(tex_c) if need_t else None
        with on_device_of(v):
            rc = getattr(_lib.lib(), "sr_rasterize_grad_" + suf)(
                b, nv, h, w, 0, int(bool(ctx.perspective)), _lib.ptr(v), _lib.ptr(tex_c), c,
                _lib.ptr(ind), _lib.ptr(coeff), _lib.ptr(go), _lib.ptr(grad_v), _lib.ptr(grad_t),
                abs(float(ctx.eps)), stream_of(v))
        _lib.check(rc, "sr_rasterize_grad")
        return grad_v, grad_t, None, None, None, None, None


def rasterize(v, tex, tri, h=256, w=0, perspective=False, eps=1e-6):
    return Rasterize.apply(v, tex, tri, h, w, perspective, eps)
