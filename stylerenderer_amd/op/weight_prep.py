"""Weight preparation for the MFMA convolutions — host side of csrc/weight_prep.hip.

    weight_prep(weight [.., Co, Ci, k, k], scale, want_sq) -> (wt [k*k, Ci, Co], wsq [Ci, Co] | None)
    adjoint(wt [k*k, C, N], flip) -> [k*k, N, C]

`wt` is the tap-major layout k_conv_mfma streams (row pitch padded to 16 bytes; the returned tensor is
a view of the padded buffer), `wsq` the demodulation matrix sum_taps (scale*W)^2 of reference
layers.py:298-300 rewritten as rsqrt(style^2 @ wsq + eps).  One launch replaces the mul / pow / sum /
permute+copy sequence per layer and step, one more its whole backward.  Both are differentiable to
any order: the second-order formulas (path-length regulariser only) are plain tensor algebra.
"""
import weakref

import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import hold_for_capture, on_device_of, require_f32, stream_of


def _pitch(n):
    return (n + 3) // 4 * 4


def _as3(weight):
    co, ci, kh, kw = weight.shape[-4:]
    if kh != kw or kh not in (1, 3) or weight.numel() != co * ci * kh * kw:
        raise RuntimeError("weight_prep: expected [Co, Ci, k, k] with k in (1, 3), got %s" % (tuple(weight.shape),))
    return co, ci, kh


class _WPrep(Function):
    @staticmethod
    def forward(ctx, weight, scale, want_sq):
        require_f32(weight, "weight_prep")
        co, ci, k = _as3(weight)
        w = weight.contiguous()
        ld = _pitch(co)
        store = torch.empty((k * k, ci, ld), dtype=w.dtype, device=w.device)
        wsq = torch.empty((ci, co) if want_sq else (0,), dtype=w.dtype, device=w.device)
        with on_device_of(w):
            rc = _lib.lib().sr_weight_prep(_lib.ptr(store), _lib.ptr(wsq), _lib.ptr(w), float(scale), co, ci, k,
                                           ld, stream_of(w))
        _lib.check(rc, "sr_weight_prep")
        ctx.save_for_backward(weight)
        ctx.set_materialize_grads(False)
        ctx.scale, ctx.want_sq = float(scale), bool(want_sq)
        if not want_sq:
            ctx.mark_non_differentiable(wsq)
        return (store if ld == co else store[:, :, :co]), wsq

    @staticmethod
    def backward(ctx, gwt, gwsq):
        (weight,) = ctx.saved_tensors
        if not ctx.want_sq:
            gwsq = None
        return _WPrepBwd.apply(gwt, gwsq, weight, ctx.scale), None, None


class _WPrepBwd(Function):
    @staticmethod
    def forward(ctx, gwt, gwsq, weight, scale):
        co, ci, k = _as3(weight)
        w = weight.contiguous()
        if gwt is None and gwsq is None:
            return torch.zeros_like(weight)
        if gwt is not None:
            gwt = gwt.contiguous()
        if gwsq is not None:
            gwsq = gwsq.contiguous()
        gw = None
        slot = getattr(weight, "_sr_grad_slot", None)
        if slot is not None and not torch.is_grad_enabled() and w is weight:
            # the parameter's slot of a flat gradient buffer (distributed.BucketedGradReducer): this kernel writes the
            # whole gradient, so it can write it THERE — no copy into the buffer afterwards (first contribution of a
            # backward only; a recorded pass keeps its own tensor)
            gw = slot()
            if gw is not None and (gw.shape != w.shape or gw.dtype != w.dtype or gw.device != w.device
                                   or not gw.is_contiguous()):
                gw = None
        if gw is None:
            gw = torch.empty_like(w)
        with on_device_of(w):
            rc = _lib.lib().sr_weight_prep_bwd(_lib.ptr(gw), _lib.ptr(gwt), _lib.ptr(gwsq), _lib.ptr(w),
                                               float(scale), co, ci, k, co, stream_of(w))
        _lib.check(rc, "sr_weight_prep_bwd")
        ctx.save_for_backward(gwsq, weight)
        ctx.scale = float(scale)
        ctx.has = (gwt is not None, gwsq is not None)
        return gw.view(weight.shape)

    @staticmethod
    def backward(ctx, gg):
        # gw = scale * gwt^T + 2 scale^2 * w * gwsq  — linear in each argument
        gwsq, weight = ctx.saved_tensors
        co, ci, k = _as3(weight)
        sc = ctx.scale
        g3 = gg.reshape(co, ci, k * k)
        d_gwt = d_gwsq = d_w = None
        if ctx.has[0] and ctx.needs_input_grad[0]:
            d_gwt = sc * g3.permute(2, 1, 0)
        if ctx.has[1]:
            w3 = weight.reshape(co, ci, k * k)
            if ctx.needs_input_grad[1]:
                d_gwsq = (2.0 * sc * sc) * (w3 * g3).sum(2).t()
            if ctx.needs_input_grad[2]:
                d_w = ((2.0 * sc * sc) * gwsq.t()[:, :, None] * g3).reshape(weight.shape)
        return d_gwt, d_gwsq, d_w, None


def weight_prep(weight, scale, want_sq=False):
    wt, wsq = _WPrep.apply(weight, scale, want_sq)
    return wt, (wsq if want_sq else None)


# ---- frozen weights (latent inversion, sampling): prepare once per weight version --------------------------------
# OPT-IN per module (`freeze_prepared_weights(net)`), never inferred from `requires_grad`: the training loop toggles
# that flag on the discriminator while its weights keep changing under the flat Adam kernel, which writes through raw
# pointers and does not bump tensor versions.
_FROZEN_ADJ = None


# networks whose parameters a replayed hipGraph rewrites (graph_train.GraphedTrainer registers its three); a WeakSet of the
# module OBJECTS, so copy.deepcopy(net) is not a member
GRAPH_WRITTEN = weakref.WeakSet()


def freeze_prepared_weights(net, flag=True):
    """Marks every convolution of `net` whose weights the caller guarantees not to change behind torch's back (in-place
    torch ops bump the version and refresh the cache): their tap-major weights, demodulation matrices and
    data-gradient adjoints are prepared once instead of per call (k_wprep / k_wadjoint: ~90 launches per inversion
    step at batch 1, where every launch is ~5 us of a 10 ms step)."""
    if flag and net in GRAPH_WRITTEN:
        raise RuntimeError("freeze_prepared_weights: this network's parameters are rewritten by a replayed hipGraph "
                           "(GraphedTrainer's EMA / optimiser graphs) — replays do not bump tensor versions, so every "
                           "version-keyed cache (prepared weights, adjoints, Winograd-domain weights) would serve stale "
                           "values.  Freeze a copy: copy.deepcopy(trainer.g_ema)")
    for m in net.modules():
        m._frozen_weights = bool(flag)          # read by weight_prep_cached through layers.*Conv2d
    return net


_BANK = None


def _bank():
    global _BANK
    if _BANK is None:
        from . import weight_bank

        _BANK = weight_bank
    return _BANK


def weight_prep_cached(module, weight, scale, want_sq=False):
    """`weight_prep` for a module: from the pass-wide batched preparation when the network's forward opened a
    `weight_bank.Scope`, from the version-keyed cache when the module was frozen with `freeze_prepared_weights`, else
    one launch."""
    if not getattr(module, "_frozen_weights", False) or weight.requires_grad:
        bank = _bank()
        hit = bank.lookup(module, want_sq) if weight is getattr(module, "weight", None) else None
        if hit is not None:
            return hit
        bank.note(module, want_sq)
        return weight_prep(weight, scale, want_sq)
    tag = (weight.data_ptr(), weight._version, str(weight.device), float(scale))
    hit = getattr(module, "_wprep_hit", None)
    if hit is None or hit[0] != tag or (want_sq and hit[2] is None):
        with torch.no_grad():
            wt, wsq = weight_prep(weight, scale, True if want_sq else False)
        wt._sr_frozen = True                       # lets ConvFn cache the adjoint of this very tensor
        hit = (tag, wt, wsq)
        module._wprep_hit = hit
    # a graph under capture bakes these addresses in: it keeps them alive even if a later weight version replaces
    # the module's entry.  (A frozen module's weights must not change while a captured graph that read them is
    # replayed: the graph would keep using the OLD prepared tensors.)
    hold_for_capture(hit)
    return hit[1], (hit[2] if want_sq else None)


def adjoint_cached(wt, flip):
    """`adjoint` of a frozen prepared weight: keyed by the tensor's address / version, the entry holds the tensor."""
    global _FROZEN_ADJ
    if _FROZEN_ADJ is None:
        from ._dispatch import DerivedCache

        _FROZEN_ADJ = DerivedCache(256)
    key = (wt.data_ptr(), wt._version, tuple(wt.shape), tuple(wt.stride()), bool(flip), str(wt.device))
    hit = _FROZEN_ADJ.get(key)
    if hit is None:
        with torch.no_grad():
            adj = _adjoint_launch(wt, flip)
            adj._sr_frozen = True              # (op.conv keeps the Winograd-domain weights of the data gradient too)
            hit = _FROZEN_ADJ.put(key, (adj, wt))
    return hit[0]


def _adjoint_launch(wt, flip):
    taps, c, n = wt.shape
    if not (wt.stride(2) == 1 and wt.stride(0) == c * wt.stride(1) and wt.stride(1) >= n):
        wt = wt.contiguous()
    ldn, ldc = wt.stride(1), _pitch(c)
    store = torch.empty((taps, n, ldc), dtype=wt.dtype, device=wt.device)
    with on_device_of(wt):
        rc = _lib.lib().sr_weight_adjoint(_lib.ptr(store), _lib.ptr(wt), taps, c, n, ldn, ldc, int(bool(flip)),
                                          stream_of(wt))
    _lib.check(rc, "sr_weight_adjoint")
    return store if ldc == c else store[:, :, :c]


class _Adjoint(Function):
    @staticmethod
    def forward(ctx, wt, flip):
        require_f32(wt, "weight adjoint")
        ctx.flip = flip
        return _adjoint_launch(wt, flip)

    @staticmethod
    def backward(ctx, g):
        return _Adjoint.apply(g, ctx.flip), None


def adjoint(wt, flip):
    """[taps, C, N] -> [taps, N, C] (taps reversed when flip): the permutation is its own inverse."""
    return _Adjoint.apply(wt, flip)
