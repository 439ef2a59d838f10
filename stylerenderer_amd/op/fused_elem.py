"""Fused element-wise passes — host side of csrc/fused_elem.hip.

  noise_bias_act(x, noise, noise_weight, bias, negative_slope, scale)
      = fused_leaky_relu(x + noise_weight * noise, bias)   (reference model.py:26-32) in ONE pass,
      twice differentiable like the reference's pair of autograd Functions.
  rowdot(a, b [, scale])  row-wise dot products (+ scaled copy) used by the modulated-conv backward.
"""
import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import mark_inputs, on_device_of, stream_of, wanted
from .fused_act import fused_leaky_relu


def _geometry(x):
    n = x.size(0)
    c = x.size(1)
    inner = 1
    for i in range(2, x.dim()):
        inner *= x.size(i)
    return n, c, inner


def _fusable(x, noise):
    if x.device.type != "cuda" or x.dtype != torch.float32 or x.dim() < 3:
        return False
    n, c, inner = _geometry(x)
    if inner % 4 != 0 or n * c > 65535 or x.data_ptr() % 16:
        return False
    if noise is not None:
        if noise.dtype != torch.float32 or noise.numel() not in (inner, n * inner) or noise.data_ptr() % 16:
            return False
    return True


def _launch_fwd(x, noise, noise_w, bias, ref, slope, scale):
    n, c, inner = _geometry(x)
    y = torch.empty_like(x)
    bstride = 0 if noise is None or noise.numel() == inner else inner
    with on_device_of(x):
        rc = _lib.lib().sr_noise_bias_act(_lib.ptr(y), _lib.ptr(x), _lib.ptr(noise), _lib.ptr(noise_w),
                                          _lib.ptr(bias), _lib.ptr(ref), float(slope), float(scale), n, c,
                                          inner, bstride, stream_of(x))
    _lib.check(rc, "sr_noise_bias_act")
    return y


class _NBABackward(Function):
    @staticmethod
    def forward(ctx, gy, out, noise, slope, scale, want_params=True):
        """want_params False: bias and noise strength are frozen (sampling, inversion) — no reduction launches, empty
        gb / gnw."""
        n, c, inner = _geometry(out)
        gy = gy.contiguous()
        gx = torch.empty_like(out)
        gb = (torch.zeros if out.numel() == 0 else torch.empty)(c if want_params else 0, dtype=out.dtype,
                                                                device=out.device)
        gnw = (torch.empty if (noise is not None and out.numel() > 0) else torch.zeros)(1 if want_params else 0,
                                                                                            dtype=out.dtype,
                                                                                            device=out.device)
        L = _lib.lib()
        scratch = torch.empty(L.sr_noise_bias_act_bwd_scratch_floats(n, c, inner), dtype=out.dtype,
                              device=out.device)
        bstride = 0 if noise is None or noise.numel() == inner else inner
        with on_device_of(out):
            rc = L.sr_noise_bias_act_bwd(_lib.ptr(gx), _lib.ptr(gb) if want_params else None,
                                         _lib.ptr(gnw) if want_params else None, _lib.ptr(gy),
                                         _lib.ptr(out), _lib.ptr(noise), float(slope), float(scale), n, c,
                                         inner, bstride, _lib.ptr(scratch), stream_of(out))
        _lib.check(rc, "sr_noise_bias_act_bwd")
        ctx.save_for_backward(out, noise)
        ctx.slope, ctx.scale = slope, scale
        return gx, gb, gnw

    @staticmethod
    def backward(ctx, ggx, ggb, ggnw):
        out, noise = ctx.saved_tensors
        c = out.shape[1]
        # (frozen parameters: gb / gnw were not produced, their cotangents are absent = zero)
        ggb = ggb.contiguous() if ggb is not None and ggb.numel() == c else out.new_zeros(c)
        if noise is not None:
            ggnw = ggnw.contiguous() if ggnw is not None and ggnw.numel() == 1 else out.new_zeros(1)
        else:
            ggnw = None
        gg = _launch_fwd(ggx.contiguous(), noise, ggnw, ggb, out, ctx.slope, ctx.scale)
        return gg, None, None, None, None, None


class _NBA(Function):
    @staticmethod
    def forward(ctx, x, noise, noise_w, bias, slope, scale, given=None):
        """given: the output a fused convolution node already computed for exactly these operands (its recorded
        backward re-derives the VJP from the separate operators): no launch, and the activation mask of every order
        of differentiation comes from the ONE tensor the forward pass returned."""
        mark_inputs(ctx, x, noise, noise_w, bias, slope, scale, given)
        y = _launch_fwd(x, noise, noise_w, bias, None, slope, scale) if given is None else given.detach()
        ctx.save_for_backward(y, noise)
        ctx.slope, ctx.scale = slope, scale
        return y

    @staticmethod
    def backward(ctx, gy):
        y, noise = ctx.saved_tensors
        needs = wanted(ctx)
        want = bool(needs[3] or (noise is not None and needs[2]))
        gx, gb, gnw = _NBABackward.apply(gy, y, noise, ctx.slope, ctx.scale, want)
        return (gx, None, (gnw if noise is not None and want and needs[2] else None), (gb if want and needs[3] else None),
                None, None, None)


def noise_bias_act(x, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    """lrelu(x + noise_weight*noise + bias) * scale; `noise` [B or 1, 1, H, W] or None."""
    if noise is not None:
        noise = noise.contiguous()
    if _fusable(x, noise) and x.is_contiguous():
        return _NBA.apply(x, noise, noise_weight if noise is not None else None, bias, negative_slope, scale)
    if noise is not None:
        x = x + noise_weight * noise
    return fused_leaky_relu(x, bias, negative_slope, scale)


class _NBAAffineBackward(Function):
    """First-order backward of `_NBAAffine` as its own node; its own backward is the fused second-order pass the
    path-length regulariser needs (gradients w.r.t. gy, x and the scale plane)."""

    @staticmethod
    def forward(ctx, gy, out, x, smap2, noise, slope, scale, want_params=True):
        n, c, inner = _geometry(out)
        gy = gy.contiguous()
        gx = torch.empty_like(out)
        gmap = torch.empty((2, n) + tuple(out.shape[2:]), dtype=out.dtype, device=out.device)   # [a | s] planes
        # want_params False: bias and noise strength are frozen (inversion, sampling) — no reduction launches
        gb = (torch.zeros if out.numel() == 0 else torch.empty)(c if want_params else 0, dtype=out.dtype,
                                                                device=out.device)
        gnw = (torch.empty if (noise is not None and out.numel() > 0) else torch.zeros)(1 if want_params else 0,
                                                                                            dtype=out.dtype,
                                                                                            device=out.device)
        L = _lib.lib()
        scratch = torch.empty(L.sr_noise_bias_act_affine_bwd_scratch_floats(n, c, inner), dtype=out.dtype,
                              device=out.device)
        bstride = 0 if noise is None or noise.numel() == inner else inner
        with on_device_of(out):
            rc = L.sr_noise_bias_act_affine_bwd(
                _lib.ptr(gx), gmap.data_ptr(), gmap.data_ptr() + 4 * n * inner, _lib.ptr(gb) if want_params else None,
                _lib.ptr(gnw) if want_params else None,
                _lib.ptr(gy), _lib.ptr(out), _lib.ptr(x), _lib.ptr(smap2), smap2.stride(0), _lib.ptr(noise),
                float(slope), float(scale), n, c, inner, bstride, _lib.ptr(scratch), stream_of(out))
        _lib.check(rc, "sr_noise_bias_act_affine_bwd")
        ctx.save_for_backward(gy, out, x, smap2, noise)
        # absent cotangents (frozen bias / noise strength, a plane nobody differentiated) arrive as None, not as
        # zero-filled tensors: two fill launches less per layer of the recorded backward
        ctx.set_materialize_grads(False)
        ctx.cfg = (float(slope), float(scale))
        return gx, gmap.transpose(0, 1), gb, gnw

    @staticmethod
    def backward(ctx, G_gx, G_gmap, G_gb, G_gnw):
        """The second-order pass (csrc/fused_elem.hip k_nba_aff_bwd2): gradients w.r.t. gy, x and the scale plane."""
        gy, out, x, smap2, noise = ctx.saved_tensors
        slope, scale = ctx.cfg
        n, c, inner = _geometry(out)
        d_gy = torch.empty_like(out)
        d_x = torch.empty_like(out)
        d_smap = torch.zeros_like(smap2, memory_format=torch.contiguous_format)       # channel 1 (shift plane): zero
        G_gx = G_gx.contiguous() if G_gx is not None else None
        G_gmap = G_gmap.contiguous() if G_gmap is not None else None                  # [n, 2, H, W]
        G_gb = G_gb.contiguous() if (G_gb is not None and G_gb.numel() == c) else None
        G_gnw = G_gnw.contiguous() if (G_gnw is not None and G_gnw.numel() == 1 and noise is not None) else None
        L = _lib.lib()
        scratch = torch.empty(L.sr_noise_bias_act_affine_bwd2_scratch_floats(n, c, inner), dtype=out.dtype,
                              device=out.device)
        bstride = 0 if noise is None or noise.numel() == inner else inner
        with on_device_of(out):
            rc = L.sr_noise_bias_act_affine_bwd2(
                _lib.ptr(d_gy), _lib.ptr(d_x), _lib.ptr(d_smap), 2 * inner, _lib.ptr(G_gx), _lib.ptr(G_gmap), 2 * inner,
                _lib.ptr(G_gb), _lib.ptr(G_gnw), _lib.ptr(gy), _lib.ptr(out), _lib.ptr(x), _lib.ptr(smap2),
                smap2.stride(0), _lib.ptr(noise), float(slope), float(scale), n, c, inner, bstride, _lib.ptr(scratch),
                stream_of(out))
        _lib.check(rc, "sr_noise_bias_act_affine_bwd2")
        needs = ctx.needs_input_grad
        return (d_gy if needs[0] else None, None, d_x if needs[2] else None, d_smap if needs[3] else None, None,
                None, None, None)


class _NBAAffine(Function):
    """y = lrelu((x * a + s) + noise_w * noise + bias) * scale with a = smap2[:, 0:1], s = smap2[:, 1:2]:
    the tail of StyledMapConv (reference model.py:49-54) in one pass."""

    @staticmethod
    def forward(ctx, x, smap2, noise, noise_w, bias, slope, scale):
        mark_inputs(ctx, x, smap2, noise, noise_w, bias, slope, scale)
        n, c, inner = _geometry(x)
        y = torch.empty_like(x)
        bstride = 0 if noise is None or noise.numel() == inner else inner
        with on_device_of(x):
            rc = _lib.lib().sr_noise_bias_act_affine(
                _lib.ptr(y), _lib.ptr(x), smap2.data_ptr(), smap2.data_ptr() + 4 * inner, smap2.stride(0),
                _lib.ptr(noise), _lib.ptr(noise_w), _lib.ptr(bias), float(slope), float(scale), n, c, inner, bstride,
                stream_of(x))
        _lib.check(rc, "sr_noise_bias_act_affine")
        ctx.save_for_backward(x, smap2, noise, noise_w, bias, y)
        ctx.cfg = (slope, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, smap2, noise, noise_w, bias, y = ctx.saved_tensors
        slope, scale = ctx.cfg
        needs = wanted(ctx)
        # (also when this backward is itself recorded — path-length regulariser: _NBAAffineBackward has a native
        # second-order pass, k_nba_aff_bwd2; round 2 re-derived the VJP from ~45 tensor-algebra launches per layer)
        want = bool(needs[4] or (noise is not None and needs[3]))
        gx, gmap, gb, gnw = _NBAAffineBackward.apply(gy, y, x, smap2, noise, slope, scale, want)
        return (gx if needs[0] else None, gmap if needs[1] else None, None,
                gnw if (noise is not None and want and needs[3]) else None, gb if (want and needs[4]) else None, None,
                None)


def noise_bias_act_affine(x, smap2, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    """lrelu((x * smap2[:, :1] + smap2[:, 1:2]) + noise_weight * noise + bias) * scale; `smap2` [B, 2, H, W]
    (a channel slice of a contiguous map tensor is fine)."""
    if noise is not None:
        noise = noise.contiguous()
    n, c, inner = _geometry(x) if x.dim() >= 3 else (0, 0, 0)
    ok = (_fusable(x, noise) and x.is_contiguous() and bias is not None and smap2.dim() == 4
          and tuple(smap2.shape) == (n, 2) + tuple(x.shape[2:]) and smap2.dtype == torch.float32
          and smap2.stride()[1:] == (inner, x.shape[3], 1) and smap2.stride(0) % 4 == 0
          and smap2.data_ptr() % 16 == 0 and n <= 65535)
    if not ok:
        out = x * smap2[:, :1] + smap2[:, 1:2]
        return noise_bias_act(out, noise, noise_weight, bias, negative_slope, scale)
    return _NBAAffine.apply(x, smap2, noise, noise_weight if noise is not None else None, bias, negative_slope, scale)


class _BlurNBA(Function):
    """Blur (4x4 FIR, up = down = 1) + noise + bias + LeakyReLU in ONE pass (csrc/upfirdn2d.hip
    k_fir4_tile<true>): the tail of an upsampling StyledConv (reference layers.py:304-311 blur after the
    transposed conv, model.py:26-32).  Its backward is the composition of the two existing differentiable
    backward operators, so gradients of any order stay on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, kernel, pad, noise, noise_w, bias, slope, scale, given=None):
        """given: see _NBA.forward."""
        n, c, ih, iw = x.shape
        p0, p1 = pad
        oh, ow = ih + p0 + p1 - 3, iw + p0 + p1 - 3
        if given is not None:
            y = given.detach()
        else:
            x = x.contiguous()
            k = kernel.contiguous()
            y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device)
            bstride = 0 if noise is None or noise.numel() == oh * ow else oh * ow
            with on_device_of(x):
                rc = _lib.lib().sr_blur_noise_bias_act(_lib.ptr(y), _lib.ptr(x), _lib.ptr(k), _lib.ptr(noise),
                                                       _lib.ptr(noise_w), _lib.ptr(bias), float(slope), float(scale), n,
                                                       c, ih, iw, oh, ow, p0, p1, bstride, stream_of(x))
            _lib.check(rc, "sr_blur_noise_bias_act")
        ctx.save_for_backward(y, noise, kernel)
        ctx.cfg = (slope, scale, p0, p1, tuple(x.shape), (oh, ow))
        return y

    @staticmethod
    def backward(ctx, gy):
        from .upfirdn2d import UpFirDn2dBackward, flipped

        y, noise, kernel = ctx.saved_tensors
        slope, scale, p0, p1, in_size, out_size = ctx.cfg
        gpre, gb, gnw = _NBABackward.apply(gy, y, noise, slope, scale)
        g_pad = (3 - p0, in_size[3] - out_size[1] + p0, 3 - p0, in_size[2] - out_size[0] + p0)
        gx = UpFirDn2dBackward.apply(gpre, kernel, flipped(kernel), (1, 1), (1, 1), (p0, p1, p0, p1),
                                     g_pad, in_size, out_size)
        return gx, None, None, None, (gnw if noise is not None else None), gb, None, None, None


def blur_noise_bias_act(x, kernel, pad, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    """lrelu(blur(x) + noise_weight*noise + bias) * scale for the 4x4 blur of the upsampling layers; falls
    back to the two separate operators for anything else."""
    from .upfirdn2d import upfirdn2d

    ok = (x.device.type == "cuda" and x.dtype == torch.float32 and x.dim() == 4 and tuple(kernel.shape) == (4, 4)
          and kernel.dtype == torch.float32 and bias is not None and x.numel() > 0)
    if noise is not None:
        noise = noise.contiguous()
        oh, ow = x.size(2) + pad[0] + pad[1] - 3, x.size(3) + pad[0] + pad[1] - 3
        ok = ok and noise.dtype == torch.float32 and noise.numel() in (oh * ow, x.size(0) * oh * ow)
    if not ok:
        return noise_bias_act(upfirdn2d(x, kernel, pad=pad), noise, noise_weight, bias, negative_slope, scale)
    k = kernel if kernel.device == x.device else kernel.to(x.device)
    return _BlurNBA.apply(x, k, tuple(pad), noise, noise_weight if noise is not None else None, bias,
                          negative_slope, scale)


def _rowdot_launch(a, b, scale):
    rows = a.size(0) * a.size(1)
    inner = a.numel() // max(rows, 1)
    dots = torch.empty(a.shape[:2], dtype=a.dtype, device=a.device)
    out = torch.empty_like(b) if scale is not None else None
    L = _lib.lib()
    scratch = torch.empty(L.sr_rowdot_scratch_floats(rows, inner), dtype=a.dtype, device=a.device)
    with on_device_of(a):
        rc = L.sr_rowdot(_lib.ptr(dots), _lib.ptr(out), _lib.ptr(a), _lib.ptr(b), _lib.ptr(scale), rows,
                         inner, _lib.ptr(scratch), stream_of(a))
    _lib.check(rc, "sr_rowdot")
    return dots, out


class _RowDot(Function):
    """(a, b, scale) -> (dots [B,C], b*scale[:, :, None, None]).  Differentiable (torch ops)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        ctx.save_for_backward(a, b, scale)
        dots, out = _rowdot_launch(a, b, scale)
        if out is None:
            return dots
        return dots, out

    @staticmethod
    def backward(ctx, gd, go=None):
        a, b, scale = ctx.saved_tensors
        if not torch.is_grad_enabled() and a.is_contiguous() and b.is_contiguous():
            # one pass (csrc/fused_elem.hip k_rowdot_bwd) instead of five element-wise launches and a reduction
            need_a, need_b, need_s = ctx.needs_input_grad[:3]
            go_c = go.contiguous() if (go is not None and scale is not None) else None
            gd_c = gd.contiguous() if gd is not None else None
            rows = a.size(0) * a.size(1)
            inner = a.numel() // max(rows, 1)
            ga = torch.empty_like(a) if need_a else None
            gb = torch.empty_like(b) if need_b else None
            gs = torch.empty(a.shape[:2], dtype=a.dtype, device=a.device) if (need_s and go_c is not None) else None
            L = _lib.lib()
            scratch = (torch.empty(L.sr_rowdot_scratch_floats(rows, inner), dtype=a.dtype, device=a.device)
                       if gs is not None else None)
            with on_device_of(a):
                rc = L.sr_rowdot_bwd(_lib.ptr(ga), _lib.ptr(gb), _lib.ptr(gs), _lib.ptr(a), _lib.ptr(b), _lib.ptr(gd_c),
                                     _lib.ptr(go_c), _lib.ptr(scale) if go_c is not None else None, rows, inner,
                                     _lib.ptr(scratch), stream_of(a))
            _lib.check(rc, "sr_rowdot_bwd")
            return ga, gb, gs
        ga = gd[:, :, None, None] * b
        gb = gd[:, :, None, None] * a
        gs = None
        if scale is not None and go is not None:
            gb = gb + go * scale[:, :, None, None]
            gs = (go * b).sum((2, 3))
        return ga, gb, gs


def rowdot(a, b, scale=None):
    """dots[b,c] = sum_hw a*b (and, with `scale` [B,C], also b*scale broadcast over hw)."""
    ok = (a.device.type == "cuda" and a.dtype == torch.float32 and a.dim() == 4 and a.shape == b.shape
          and a.size(0) * a.size(1) <= 65535 and a.is_contiguous() and b.is_contiguous())
    if not ok:
        dots = (a * b).sum((2, 3))
        return dots if scale is None else (dots, b * scale[:, :, None, None])
    return _RowDot.apply(a, b, scale.contiguous() if scale is not None else None)


def rowdot_div(a, b, div):
    """rowdot(a, b) / div [B, C].  Outside a recorded pass (no autograd graph is being built) one kernel does both
    (sr_rowdot_div); a recorded pass composes the two differentiable operators."""
    ok = (not torch.is_grad_enabled() and a.device.type == "cuda" and a.dtype == torch.float32 and a.dim() == 4
          and a.shape == b.shape and a.size(0) * a.size(1) <= 65535 and a.is_contiguous() and b.is_contiguous()
          and div.dtype == torch.float32 and tuple(div.shape) == tuple(a.shape[:2]) and div.is_contiguous())
    if not ok:
        return rowdot(a, b) / div
    rows = a.size(0) * a.size(1)
    inner = a.numel() // max(rows, 1)
    dots = torch.empty(a.shape[:2], dtype=a.dtype, device=a.device)
    L = _lib.lib()
    scratch = torch.empty(L.sr_rowdot_scratch_floats(rows, inner), dtype=a.dtype, device=a.device)
    with on_device_of(a):
        rc = L.sr_rowdot_div(_lib.ptr(dots), _lib.ptr(a), _lib.ptr(b), _lib.ptr(div), rows, inner, _lib.ptr(scratch),
                             stream_of(a))
    _lib.check(rc, "sr_rowdot_div")
    return dots
