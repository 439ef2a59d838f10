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
attribute interpolation happens in the resolve pass (index / coeff are not even written: the backward
walks a 4-byte winner map), and the backward is a deterministic two-phase gather into grad_v / grad_tex
instead of a COO matrix + sparse.mm per call (reference op/rasterize.py:46-77).

CPU tensors take the library's host loops (sr_rasterize_*_cpu_*: the reference's extension also serves
CPU tensors, op/rasterize.cpp:126-150) with the interpolation / scatter in torch like the reference's
Python; that branch is keyed on the tensors' device and never reached for `cuda` tensors.
"""
import os
import types

import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import DerivedCache, is_device_tensor, on_device_of, stream_of


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


SR_RASTER_GRAD_ACC = 4     # sr_rasterize_grad_*: add to what grad_v / grad_tex hold (RasterizePyramid)
SR_RASTER_CHW = 2          # include/stylerenderer_amd.h: attribute maps / their gradient channel-major [b, c, h, w]


def _forward_impl(vertices, triangles, height, width, perspective, eps, tex=None, want_z=False,
                  want_index=True, want_win=False, chw=False):
    if vertices.device != triangles.device:
        raise RuntimeError(" cuda input error")
    if triangles.dtype != torch.int64:
        raise RuntimeError(" type error")
    suf = _suffix(vertices)
    b, nv, nf, rv, rf = _geometry(vertices, triangles)
    h = 1 if height <= 0 else int(height)
    w = h if width <= 0 else int(width)
    dev, dt = vertices.device, vertices.dtype
    L = _lib.lib()
    if not is_device_tensor(vertices):
        # host loops of the library (reference rasterize_cpu, op/rasterize.cpp:21-67)
        index = torch.empty((b, h, w, 3), dtype=torch.int64)
        coeff = torch.empty((b, h, w, 3), dtype=dt)
        zbuf = torch.empty((b, h, w), dtype=dt) if want_z else None
        rc = getattr(L, "sr_rasterize_forward_cpu_" + suf)(
            b, nv, nf, h, w, int(rv), int(rf), int(bool(perspective)), _lib.ptr(vertices), _lib.ptr(triangles),
            _lib.ptr(index), _lib.ptr(coeff), _lib.ptr(zbuf), abs(float(eps)))
        _lib.check(rc, "sr_rasterize_forward_cpu")
        attr = None
        if tex is not None:
            tex_c = 1 if tex.dim() == vertices.dim() - 1 else int(tex.shape[-1])
            flat = tex.contiguous().view(-1, tex_c)
            if flat.dtype != dt:
                raise RuntimeError(" type error")
            taken = flat[index.view(-1)].view(b, h, w, 3, tex_c) * coeff.unsqueeze(-1)
            attr = (taken[..., 0, :] + taken[..., 1, :]) + taken[..., 2, :]
            if chw:
                attr = attr.permute(0, 3, 1, 2).contiguous()
        if rv and rf:
            return index[0], coeff[0], (zbuf[0] if zbuf is not None else None), (attr[0] if attr is not None else None), None
        return index, coeff, zbuf, attr, None
    index = torch.empty((b, h, w, 3), dtype=torch.int64, device=dev) if want_index else None
    coeff = torch.empty((b, h, w, 3), dtype=dt, device=dev) if want_index else None
    zbuf = torch.empty((b, h, w), dtype=dt, device=dev) if want_z else None
    win = torch.empty((b, h, w), dtype=torch.int32, device=dev) if want_win else None
    # gradient state: [count | big-triangle ids b*nf | leader table b*nf | state word] (include/stylerenderer_amd.h)
    big = torch.empty(2 + 2 * b * nf, dtype=torch.int32, device=dev) if want_win else None
    attr, tex_c, tex_flat = None, 0, None
    if tex is not None:
        tex_c = 1 if tex.dim() == vertices.dim() - 1 else int(tex.shape[-1])
        tex_flat = tex.contiguous().view(-1, tex_c)
        if tex_flat.dtype != dt:
            raise RuntimeError(" type error")
        attr = torch.empty((b, tex_c, h, w) if chw else (b, h, w, tex_c), dtype=dt, device=dev)
    work = torch.empty(L.sr_rasterize_scratch_bytes(b, nf, h, w, int(suf == "f64")), dtype=torch.uint8,
                       device=dev)
    with on_device_of(vertices):
        rc = getattr(L, "sr_rasterize_forward_" + suf)(
            b, nv, nf, h, w, int(rv), int(rf), int(bool(perspective)) | (SR_RASTER_CHW if chw else 0), _lib.ptr(vertices),
            _lib.ptr(triangles), _lib.ptr(index), _lib.ptr(coeff), _lib.ptr(zbuf), abs(float(eps)),
            _lib.ptr(tex_flat), tex_c, _lib.ptr(attr), _lib.ptr(win), _lib.ptr(big), _lib.ptr(work),
            stream_of(vertices))
    _lib.check(rc, "sr_rasterize_forward")
    if rv and rf:
        index = index[0] if index is not None else None
        coeff = coeff[0] if coeff is not None else None
        zbuf = zbuf[0] if zbuf is not None else None
        attr = attr[0] if attr is not None else None
    return index, coeff, zbuf, attr, ((win, big) if want_win else None)


def forward(vertices, triangles, height, width, perspective=False, eps=1e-9):
    index, coeff, _, _, _ = _forward_impl(vertices, triangles, height, width, perspective, eps)
    return [index, coeff]


def forward_with_depth(vertices, triangles, height, width, perspective=False, eps=1e-9):
    """Extension used by tests: also returns the z-buffer the reference keeps internal."""
    index, coeff, zbuf, _, _ = _forward_impl(vertices, triangles, height, width, perspective, eps,
                                             want_z=True)
    return index, coeff, zbuf


def backward(vertices, index, perspective=False, eps=1e-9):
    if vertices.device != index.device:
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
    if not is_device_tensor(vertices):
        rc = getattr(_lib.lib(), "sr_rasterize_backward_cpu_" + suf)(
            b, n, h, w, int(bool(perspective)), _lib.ptr(vertices), _lib.ptr(index), _lib.ptr(dcoeff),
            abs(float(eps)))
        _lib.check(rc, "sr_rasterize_backward_cpu")
        return dcoeff
    with on_device_of(vertices):
        rc = getattr(_lib.lib(), "sr_rasterize_backward_" + suf)(
            b, n, h, w, int(rv), int(bool(perspective)), _lib.ptr(vertices), _lib.ptr(index),
            _lib.ptr(dcoeff), abs(float(eps)), stream_of(vertices))
    _lib.check(rc, "sr_rasterize_backward")
    return dcoeff


# the reference's extension module object, by name
rasterize_op = types.SimpleNamespace(forward=forward, backward=backward)


# ---- per-topology incidence list for the gradient gather ------------------------------------------
_INC_CACHE = DerivedCache(16)


def _incidence_one(tri, nv):
    """CSR incidence of one [nf, 3] triangle list: (off int32 [nv + 2], adj int32 [3 nf]); the entries of vertex
    i are the ascending corner-major indices k * nf + f with tri[f, k] == i.  Out-of-range ids (triangles the
    rasterizer skips, reference op/rasterize.cpp:30-33) are filed under the extra bucket nv."""
    flat = tri.t().reshape(-1)
    flat = torch.where((flat < 0) | (flat >= nv), torch.full_like(flat, nv), flat)
    order64 = torch.sort(flat, stable=True)[1]
    order = order64.to(torch.int32)
    counts = torch.bincount(flat, minlength=nv + 1)
    off = torch.zeros(nv + 2, dtype=torch.int32, device=tri.device)
    off[1:] = torch.cumsum(counts, 0).to(torch.int32)
    # the inverse permutation: corner k * nf + f sits at list position slot[k * nf + f] — the vertex-major slot the
    # gradient's first phase writes that corner's values to (include/stylerenderer_amd.h, sr_rasterize_grad_*)
    slot = torch.empty_like(order)
    slot[order64] = torch.arange(order.numel(), dtype=torch.int32, device=tri.device)
    return off, order.contiguous(), slot


def incidence(tri, nv):
    """(adj_off, adj, off_bstride, adj_bstride, adj_slot) for tri [nf, 3] (shared) or [b, nf, 3]; cached per tensor."""
    key = (tri.data_ptr(), tuple(tri.shape), tri._version, str(tri.device), int(nv))
    hit = _INC_CACHE.get(key)
    if hit is not None:
        return hit[:5]
    if tri.dim() == 2:
        off, adj, slot = _incidence_one(tri, nv)
        val = (off, adj, 0, 0, slot, tri)
    else:
        parts = [_incidence_one(t, nv) for t in tri]
        off = torch.stack([p[0] for p in parts]).contiguous()
        adj = torch.stack([p[1] for p in parts]).contiguous()
        slot = torch.stack([p[2] for p in parts]).contiguous()
        val = (off, adj, off.shape[1], adj.shape[1], slot, tri)
    _INC_CACHE.put(key, val)                                 # holds `tri`: the key is its address
    return val[:5]


def _device_grad(v, tex, tri, win, big, grad_out, chw, perspective, eps, no_channel, need_v, need_t, into=None):
    """One sr_rasterize_grad_* call over the gradient state (win, big) of a device forward.  `into` = (grad_v, grad_t) of
    an earlier call over the same mesh: this call ADDS to them (SR_RASTER_GRAD_ACC)."""
    if v.dim() != 3:
        raise RuntimeError("rasterize backward: batched vertices [b, n, 3] required")
    if win is None:
        raise RuntimeError("rasterize backward: the forward pass recorded no gradient state")
    suf = _suffix(v)
    b, nv = v.size(0), v.size(1)
    nf = tri.size(-2)
    h, w = win.shape[-2], win.shape[-1]
    c = 1 if no_channel else int(tex.shape[-1])
    go = grad_out.contiguous()                            # [b, h, w, c], or [b, c, h, w] for a chw forward
    tex_c = tex.contiguous()
    if tex_c.dim() < 2 or tex_c.shape[0] != b or tex_c.shape[1] != nv:
        raise RuntimeError("rasterize backward: batched attributes [b, n(, c)] required")
    if into is not None:
        grad_v, grad_t = into
    else:
        grad_v = torch.empty_like(v) if need_v else None
        grad_t = torch.empty_like(tex_c) if need_t else None
    off, adj, off_bs, adj_bs, slot = incidence(tri, nv)
    L = _lib.lib()
    work = torch.empty(L.sr_rasterize_grad_scratch_bytes(b, nf, c, int(suf == "f64")), dtype=torch.uint8, device=v.device)
    flags = int(bool(perspective)) | (SR_RASTER_CHW if chw else 0) | (SR_RASTER_GRAD_ACC if into is not None else 0)
    with on_device_of(v):
        rc = getattr(L, "sr_rasterize_grad_" + suf)(
            b, nv, nf, h, w, int(tri.dim() == 2), flags, _lib.ptr(v), _lib.ptr(tex_c), c,
            _lib.ptr(tri), _lib.ptr(win), _lib.ptr(big), _lib.ptr(go), _lib.ptr(off), _lib.ptr(adj), off_bs, adj_bs,
            _lib.ptr(slot), _lib.ptr(grad_v), _lib.ptr(grad_t), abs(float(eps)), _lib.ptr(work), stream_of(v))
    _lib.check(rc, "sr_rasterize_grad")
    return grad_v, grad_t


class Rasterize(Function):
    @staticmethod
    def forward(ctx, v, tex, tri, h, w, perspective, eps, chw=False):
        """chw: the interpolated attributes come back channel-major [b, c, h, w] (and the gradient is taken in that
        layout) — the layout the generator's map heads convolve; same values as the reference's [b, h, w, c]."""
        v = v.contiguous()
        tri = tri.contiguous()
        on_dev = is_device_tensor(v)
        need_grad = on_dev and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        chw = bool(chw) and tex.dim() == v.dim()
        ind, coeff, _, out, state = _forward_impl(v, tri, h, w, perspective, eps, tex=tex,
                                                  want_index=not on_dev, want_win=need_grad, chw=chw)
        ctx.chw = chw
        if on_dev:
            win, big = state if state is not None else (None, None)
            ctx.save_for_backward(v, tex, tri, win, big)
        else:
            ctx.save_for_backward(v, tex, tri, ind, coeff)
        ctx.on_dev = on_dev
        ctx.perspective = perspective
        ctx.eps = eps
        ctx.no_channel = tex.dim() == v.dim() - 1
        return out[..., 0] if ctx.no_channel else out

    @staticmethod
    def backward(ctx, grad_out):
        need_v, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_v or need_t):
            return (None,) * 8
        if not ctx.on_dev:
            if ctx.chw:
                grad_out = grad_out.permute(0, 2, 3, 1)
            return Rasterize._backward_host(ctx, grad_out, need_v, need_t) + (None,)
        v, tex, tri, win, big = ctx.saved_tensors
        grad_v, grad_t = _device_grad(v, tex, tri, win, big, grad_out, ctx.chw, ctx.perspective, ctx.eps, ctx.no_channel,
                                      need_v, need_t)
        return grad_v, grad_t, None, None, None, None, None, None

    @staticmethod
    def _backward_host(ctx, grad_out, need_v, need_t):
        """CPU tensors: dcoeff from the host loop, then the reference's algebra (op/rasterize.py:46-77) with
        index_add in place of the COO sparse matmul."""
        v, tex, tri, ind, coeff = ctx.saved_tensors
        if v.dim() != 3:
            raise RuntimeError("rasterize backward: batched vertices [b, n, 3] required")
        b, nv = v.size(0), v.size(1)
        c = 1 if ctx.no_channel else int(tex.shape[-1])
        go = grad_out.reshape(tuple(ind.shape[:-1]) + (c,))
        flat_idx = ind.reshape(-1)
        grad_v = grad_t = None
        if need_v:
            dcoeff = backward(v, ind, ctx.perspective, ctx.eps)                     # [b,h,w,3,9]
            tex_rast = tex.reshape(-1, c)[flat_idx].view(tuple(ind.shape) + (c,))   # [b,h,w,3,c]
            dl_dw = (go.unsqueeze(-2) * tex_rast).sum(-1)                           # [b,h,w,3]
            diff = torch.matmul(dl_dw.unsqueeze(-2), dcoeff).reshape(-1, 3)         # [b*h*w*3, 3]
            grad_v = torch.zeros(b * nv, 3, dtype=v.dtype).index_add_(0, flat_idx, diff).view_as(v)
        if need_t:
            contrib = (go.unsqueeze(-2) * coeff.unsqueeze(-1)).reshape(-1, c)
            grad_t = torch.zeros(b * nv, c, dtype=tex.dtype).index_add_(0, flat_idx, contrib).view_as(tex)
        return grad_v, grad_t, None, None, None, None, None


def rasterize(v, tex, tri, h=256, w=0, perspective=False, eps=1e-6, channel_major=False):
    """reference op/rasterize.py `rasterize`; channel_major=True returns the interpolated attributes as [b, c, h, w]
    (= the reference's output .permute(0, 3, 1, 2), contiguous) straight from the kernel."""
    return Rasterize.apply(v, tex, tri, h, w, perspective, eps, channel_major)


def _forward_levels(v, tex, tri, sizes, perspective, eps, want_win, chw):
    """All resolutions of `sizes` in three launches (sr_rasterize_forward_levels_f32): (attribute maps, [win, big] per
    level), or (None, None) when the call is outside that entry point (fp64, shared vertices, a level the dispatcher
    would give to the LDS-tiled path, SR_RASTER_LEVELS=0)."""
    import ctypes

    if (v.dtype != torch.float32 or tex.dtype != torch.float32 or tri.dtype != torch.int64 or v.dim() != 3
            or tex.dim() != 3 or os.environ.get("SR_RASTER_LEVELS", "1") == "0"):
        return None, None
    b, nv, nf, rv, rf = _geometry(v, tri)
    n = len(sizes)
    L = _lib.lib()
    hs = (ctypes.c_int64 * n)(*[h for h, _ in sizes])
    ws = (ctypes.c_int64 * n)(*[w for _, w in sizes])
    if rv or not L.sr_rasterize_levels_supported(n, b, nf, hs, ws):
        return None, None
    dev = v.device
    tex_c = int(tex.shape[-1])
    tex_flat = tex.contiguous().view(-1, tex_c)
    attrs = [torch.empty((b, tex_c, h, w) if chw else (b, h, w, tex_c), dtype=v.dtype, device=dev) for h, w in sizes]
    wins = [torch.empty((b, h, w), dtype=torch.int32, device=dev) if want_win else None for h, w in sizes]
    bigs = [torch.empty(2 + 2 * b * nf, dtype=torch.int32, device=dev) if want_win else None for _ in sizes]
    sizes_b = [int(L.sr_rasterize_scratch_bytes(b, nf, h, w, 0)) for h, w in sizes]
    offs, total = [], 0
    for sz in sizes_b:
        offs.append(total)
        total += (sz + 255) // 256 * 256
    work = torch.empty(total, dtype=torch.uint8, device=dev)
    pa = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts])      # noqa: E731
    with on_device_of(v):
        rc = L.sr_rasterize_forward_levels_f32(
            n, b, nv, nf, hs, ws, int(rv), int(rf), int(bool(perspective)) | (SR_RASTER_CHW if chw else 0), _lib.ptr(v),
            _lib.ptr(tri), abs(float(eps)), _lib.ptr(tex_flat), tex_c, pa(attrs), pa(wins) if want_win else None,
            pa(bigs) if want_win else None, (ctypes.c_void_p * n)(*[work.data_ptr() + o for o in offs]), stream_of(v))
    _lib.check(rc, "sr_rasterize_forward_levels_f32")
    states = []
    for wn, bg in zip(wins, bigs):
        states += [wn, bg]
    return attrs, states


def _grad_levels(v, tex, tri, wins, bigs, grads, chw, perspective, eps, need_v, need_t):
    """The gradients of several levels of one mesh in four launches (sr_rasterize_grad_levels_f32): (grad_v, grad_t), or
    None when the call is outside that entry point (fp64, > 4 attribute channels, SR_RASTER_LEVELS=0): the caller then
    accumulates level by level."""
    import ctypes

    if (v.dtype != torch.float32 or tex.dim() != 3 or int(tex.shape[-1]) > 4 or v.dim() != 3 or tri.size(-2) == 0
            or any(w is None or b is None for w, b in zip(wins, bigs)) or os.environ.get("SR_RASTER_LEVELS", "1") == "0"):
        return None
    n = len(grads)
    L = _lib.lib()
    b, nv = v.size(0), v.size(1)
    nf = tri.size(-2)
    c = int(tex.shape[-1])
    tex_c = tex.contiguous()
    gos = [g.contiguous() for g in grads]
    off, adj, off_bs, adj_bs, slot = incidence(tri, nv)
    grad_v = torch.empty_like(v) if need_v else None
    grad_t = torch.empty_like(tex_c) if need_t else None
    per = (int(L.sr_rasterize_grad_scratch_bytes(b, nf, c, 0)) + 255) // 256 * 256
    work = torch.empty(per * n, dtype=torch.uint8, device=v.device)
    pa = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])                                   # noqa: E731
    hs = (ctypes.c_int64 * n)(*[int(w.shape[-2]) for w in wins])
    ws = (ctypes.c_int64 * n)(*[int(w.shape[-1]) for w in wins])
    with on_device_of(v):
        rc = L.sr_rasterize_grad_levels_f32(
            n, b, nv, nf, hs, ws, int(tri.dim() == 2), int(bool(perspective)) | (SR_RASTER_CHW if chw else 0), _lib.ptr(v),
            _lib.ptr(tex_c), c, _lib.ptr(tri), pa(wins), pa(bigs), pa(gos), _lib.ptr(off), _lib.ptr(adj), off_bs, adj_bs,
            _lib.ptr(slot), _lib.ptr(grad_v), _lib.ptr(grad_t), abs(float(eps)),
            (ctypes.c_void_p * n)(*[work.data_ptr() + per * k for k in range(n)]), stream_of(v))
    _lib.check(rc, "sr_rasterize_grad_levels_f32")
    return grad_v, grad_t


class RasterizePyramid(Function):
    """The same posed mesh rasterised at several resolutions as ONE node (GeneratorWithMap draws a normal map per
    synthesis resolution: reference model.py:255-262, seven calls at 256^2).  Forward: the per-resolution launches of
    `Rasterize`, unchanged.  Backward: one gradient pass per resolution, each ADDING into the same vertex / attribute
    gradient buffers in list order (SR_RASTER_GRAD_ACC) — autograd otherwise sums the seven [b, nv, 3] pairs with twelve
    tensor additions per backward.  Device tensors, batched [b, nv, 3] vertices; same values as the separate calls up to
    the order of those seven additions (fixed: deterministic)."""

    @staticmethod
    def forward(ctx, v, tex, tri, sizes, perspective, eps, chw):
        v = v.contiguous()
        tri = tri.contiguous()
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        chw = bool(chw) and tex.dim() == v.dim()
        outs, states = _forward_levels(v, tex, tri, sizes, perspective, eps, need_grad, chw)
        if outs is None:
            outs, states = [], []
            for h, w in sizes:
                _, _, _, out, state = _forward_impl(v, tri, h, w, perspective, eps, tex=tex, want_index=False,
                                                    want_win=need_grad, chw=chw)
                outs.append(out)
                states += list(state) if state is not None else [None, None]
        ctx.cfg = (chw, perspective, eps, len(sizes))
        ctx.set_materialize_grads(False)          # a map nobody differentiated arrives as None, not as a zero-filled tensor
        ctx.save_for_backward(v, tex, tri, *states)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        need_v, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_v or need_t):
            return (None,) * 7
        chw, perspective, eps, n = ctx.cfg
        v, tex, tri = ctx.saved_tensors[:3]
        states = ctx.saved_tensors[3:]
        live = [k for k in range(n) if grads[k] is not None]
        fused = _grad_levels(v, tex, tri, [states[2 * k] for k in live], [states[2 * k + 1] for k in live],
                             [grads[k] for k in live], chw, perspective, eps, need_v, need_t) if len(live) > 1 else None
        if fused is not None:
            return fused[0], fused[1], None, None, None, None, None
        into = None
        for k in range(n):
            if grads[k] is None:
                continue
            into = _device_grad(v, tex, tri, states[2 * k], states[2 * k + 1], grads[k], chw, perspective, eps, False,
                                need_v, need_t, into=into)
        if into is None:
            return (None,) * 7
        return into[0], into[1], None, None, None, None, None


def rasterize_pyramid(v, tex, tri, sizes, perspective=False, eps=1e-6, channel_major=False):
    """[rasterize(v, tex, tri, h, w) for (h, w) in sizes] — on device tensors one autograd node whose backward sums the
    per-resolution gradients inside the gather kernels (RasterizePyramid)."""
    sizes = [(int(h), int(w)) for h, w in sizes]
    if (is_device_tensor(v) and v.dim() == 3 and tex.dim() == 3 and len(sizes) > 1
            and os.environ.get("SR_RASTER_PYRAMID", "1") != "0"):
        ctxless = RasterizePyramid.apply(v, tex, tri, tuple(sizes), perspective, eps, channel_major)
        return list(ctxless)
    return [rasterize(v, tex, tri, h, w, perspective, eps, channel_major) for h, w in sizes]
