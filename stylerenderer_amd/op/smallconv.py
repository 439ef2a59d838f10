"""1x1 modulated convolution with <= 4 output channels (ToRGB) — host side of the sr_smallconv_*
kernels in csrc/fused_elem.hip.  Three bilinear maps that are each other's derivatives, each an
autograd Function, so gradients of any order (path-length regulariser) stay on the HIP kernels:

    fwd(x, ws)  out[b,j,p] = sum_c ws[b,j,c] x[b,c,p]        d/dx -> dx(g, ws)    d/dws -> dw(g, x)
    dx(g, ws)   dx[b,c,p]  = sum_j ws[b,j,c] g[b,j,p]        d/dg -> fwd(gg, ws)  d/dws -> dw(g, gg)
    dw(g, x)    dws[b,j,c] = sum_p g[b,j,p] x[b,c,p]         d/dg -> fwd(x, gw)   d/dx  -> dx(g, gw)
"""
import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import mark_inputs, on_device_of, stream_of, wanted


def supported(x, n_out):
    return (x.device.type == "cuda" and x.dtype == torch.float32 and x.dim() == 4 and 1 <= n_out <= 4
            and (x.size(2) * x.size(3)) % 4 == 0 and x.size(0) * x.size(1) <= 65535 and x.size(1) <= 4096)


def _fwd(x, ws, bias=None):
    x, ws = x.contiguous(), ws.contiguous()
    b, c, h, w = x.shape
    n = ws.size(1)
    out = torch.empty((b, n, h, w), dtype=x.dtype, device=x.device)
    with on_device_of(x):
        rc = _lib.lib().sr_smallconv_fwd(_lib.ptr(out), _lib.ptr(x), _lib.ptr(ws),
                                         _lib.ptr(bias.contiguous()) if bias is not None else None, b, c, n, h * w,
                                         stream_of(x))
    _lib.check(rc, "sr_smallconv_fwd")
    return out


def _dx(g, ws, addend=None):
    """W^T g (+ addend [B, C, H, W]: the other gradient of the same feature map, added in the same pass)."""
    g, ws = g.contiguous(), ws.contiguous()
    b, n, h, w = g.shape
    c = ws.size(2)
    dx = torch.empty((b, c, h, w), dtype=g.dtype, device=g.device)
    with on_device_of(g):
        rc = _lib.lib().sr_smallconv_dx_add(_lib.ptr(dx), _lib.ptr(g), _lib.ptr(ws), _lib.ptr(addend), b, c, n, h * w,
                                            stream_of(g))
    _lib.check(rc, "sr_smallconv_dx_add")
    return dx


def _dw(g, x, want_bias=False):
    """dws [B, N, C] (and, with want_bias, gb [N] = sum over samples and pixels of g — summed by the same launch)."""
    g, x = g.contiguous(), x.contiguous()
    b, n, h, w = g.shape
    c = x.size(1)
    L = _lib.lib()
    dws = torch.empty((b, n, c), dtype=g.dtype, device=g.device)
    gb = torch.empty(n, dtype=g.dtype, device=g.device) if want_bias else None
    scratch = torch.empty(L.sr_smallconv_dw_scratch_floats(b, c, n, h * w), dtype=g.dtype, device=g.device)
    with on_device_of(g):
        rc = L.sr_smallconv_dw_bias(_lib.ptr(dws), _lib.ptr(gb), _lib.ptr(g), _lib.ptr(x), b, c, n, h * w,
                                    _lib.ptr(scratch), stream_of(g))
    _lib.check(rc, "sr_smallconv_dw_bias")
    return (dws, gb) if want_bias else dws


class SmallConvFwd(Function):
    @staticmethod
    def forward(ctx, x, ws, bias=None):
        mark_inputs(ctx, x, ws, bias)
        ctx.save_for_backward(x, ws)
        return _fwd(x, ws, bias)

    @staticmethod
    def backward(ctx, g):
        x, ws = ctx.saved_tensors
        needs = wanted(ctx)                      # (a pass that only asks for the latents' gradient: no bias sum)
        gx = SmallConvDx.apply(g, ws) if needs[0] else None
        want_b = len(needs) > 2 and needs[2]
        gw = gb = None
        if needs[1] and want_b:
            gw, gb = SmallConvDwBias.apply(g, x)          # the bias gradient rides in the weight-gradient launch
        elif needs[1]:
            gw = SmallConvDw.apply(g, x)
        elif want_b:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb


class SmallConvFork(Function):
    """(x, rgb) = (x handed on unchanged, conv1x1(x, ws) + bias): the feature map that feeds ToRGB also feeds the next
    layer (reference model.py:206-219).  As two consumers of one tensor, autograd adds their gradients in a
    read-read-write pass of the full map; as the two outputs of ONE node both arrive together and the addition rides
    in the data-gradient kernel (sr_smallconv_dx_add)."""

    @staticmethod
    def forward(ctx, x, ws, bias=None):
        mark_inputs(ctx, x, ws, bias)
        ctx.save_for_backward(x, ws)
        ctx.set_materialize_grads(False)                 # an unused output's cotangent stays None (no zero-fill)
        return x.view_as(x), _fwd(x, ws, bias)

    @staticmethod
    def backward(ctx, g_x, g):
        x, ws = ctx.saved_tensors
        needs = wanted(ctx)
        if g is None:
            return (g_x if needs[0] else None), None, None
        gx = None
        if needs[0]:
            fused = (g_x is not None and not torch.is_grad_enabled() and g_x.is_contiguous() and g_x.data_ptr() % 16 == 0
                     and g_x.dtype == g.dtype and g_x.shape == x.shape)
            if fused:
                gx = _dx(g, ws, g_x)
            else:
                gx = SmallConvDx.apply(g, ws)
                if g_x is not None:
                    gx = gx + g_x
        want_b = len(needs) > 2 and needs[2]
        gw = gb = None
        if needs[1] and want_b:
            gw, gb = SmallConvDwBias.apply(g, x)
        elif needs[1]:
            gw = SmallConvDw.apply(g, x)
        elif want_b:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb


class SmallConvDx(Function):
    @staticmethod
    def forward(ctx, g, ws):
        ctx.save_for_backward(g, ws)
        return _dx(g, ws)

    @staticmethod
    def backward(ctx, gg):
        g, ws = ctx.saved_tensors
        d_g = SmallConvFwd.apply(gg, ws) if ctx.needs_input_grad[0] else None
        d_ws = SmallConvDw.apply(g, gg) if ctx.needs_input_grad[1] else None
        return d_g, d_ws


class SmallConvDw(Function):
    @staticmethod
    def forward(ctx, g, x):
        ctx.save_for_backward(g, x)
        return _dw(g, x)

    @staticmethod
    def backward(ctx, gw):
        g, x = ctx.saved_tensors
        d_g = SmallConvFwd.apply(x, gw) if ctx.needs_input_grad[0] else None
        d_x = SmallConvDx.apply(g, gw) if ctx.needs_input_grad[1] else None
        return d_g, d_x


class SmallConvDwBias(Function):
    """(dws, gb) = (sum_p g x, sum_{b,p} g) in one launch; both linear in g, so the pull-back is the convolution of the
    cotangents with the bias cotangent as its bias."""

    @staticmethod
    def forward(ctx, g, x):
        ctx.save_for_backward(g, x)
        return _dw(g, x, True)

    @staticmethod
    def backward(ctx, gw, ggb):
        g, x = ctx.saved_tensors
        d_g = d_x = None
        if ctx.needs_input_grad[0]:
            if gw is None:
                d_g = ggb.view(1, -1, 1, 1).expand_as(g) if ggb is not None else None
            else:
                d_g = SmallConvFwd.apply(x, gw, ggb)
        if ctx.needs_input_grad[1] and gw is not None:
            d_x = SmallConvDx.apply(g, gw)
        return d_g, d_x


class _ModRows(Function):
    """ws[b,j,c] = (scale * w[j,c]) * s[b,c] in one launch (sr_modrows_fwd); its first-order pull-back in one launch too
    (sr_modrows_bwd) instead of the six multiply / reduce launches of the differentiated tensor products.  A RECORDED
    backward (path-length regulariser, R1) re-derives the pull-back from the defining products under autograd."""

    @staticmethod
    def forward(ctx, w, s, scale):
        mark_inputs(ctx, w, s, scale)
        w_, s_ = w.contiguous(), s.contiguous()
        n, c = w_.shape
        b = s_.shape[0]
        ws = torch.empty((b, n, c), dtype=s.dtype, device=s.device)
        with on_device_of(s):
            rc = _lib.lib().sr_modrows_fwd(_lib.ptr(ws), _lib.ptr(w_), _lib.ptr(s_), float(scale), b, n, c, stream_of(s))
        _lib.check(rc, "sr_modrows_fwd")
        ctx.save_for_backward(w, s)
        ctx.scale = float(scale)
        return ws

    @staticmethod
    def backward(ctx, g):
        w, s = ctx.saved_tensors
        need_w, need_s = wanted(ctx)[:2]
        if torch.is_grad_enabled():
            with torch.enable_grad():
                wa, sa = w.view_as(w), s.view_as(s)          # aliases: gradients stop at the node boundary
                ws = (wa * ctx.scale)[None, :, :] * sa[:, None, :]
                sel = [t for t, nd in ((wa, need_w), (sa, need_s)) if nd]
                got = iter(torch.autograd.grad(ws, sel, g, create_graph=True)) if sel else iter(())
                return (next(got) if need_w else None), (next(got) if need_s else None), None
        g_, w_, s_ = g.contiguous(), w.contiguous(), s.contiguous()
        n, c = w_.shape
        b = s_.shape[0]
        gw = torch.empty_like(w_) if need_w else None
        gs = torch.empty_like(s_) if need_s else None
        if gw is not None or gs is not None:
            with on_device_of(s):
                rc = _lib.lib().sr_modrows_bwd(_lib.ptr(gs), _lib.ptr(gw), _lib.ptr(g_), _lib.ptr(w_), _lib.ptr(s_),
                                               ctx.scale, b, n, c, stream_of(s))
            _lib.check(rc, "sr_modrows_bwd")
        return gw, gs, None


def modulated_rows(weight_nc, style, scale):
    """[B, N, C] = (scale * weight_nc [N, C]) * style [B, C] — the per-sample weights of the 1x1 convolution."""
    if (style.device.type == "cuda" and style.dtype == torch.float32 and weight_nc.dtype == torch.float32
            and weight_nc.dim() == 2 and style.dim() == 2):
        return _ModRows.apply(weight_nc, style, float(scale))
    return (weight_nc * scale)[None, :, :] * style[:, None, :]


def modulated_conv1x1_small(x, weight_jc, style, bias=None, scale=None, fork=False):
    """weight_jc [N, C], style [B, C], optional bias [N] -> [B, N, H, W].  `scale` None: weight_jc is already scaled
    (two tensor products); a number: the raw parameter view, scaled and modulated in one launch (`modulated_rows`).
    fork: returns (x', out) with x' = x for the next layer to consume — see SmallConvFork."""
    if scale is None:
        ws = weight_jc[None, :, :] * style[:, None, :]
    else:
        ws = modulated_rows(weight_jc, style, scale)
    if fork:
        return SmallConvFork.apply(x, ws, bias)
    return SmallConvFwd.apply(x, ws, bias)
