"""All modulation linears and demodulation factors of a generator pass in a handful of launches.

The reference evaluates, per modulated convolution (20 at 256^2: 13 StyledConv + 7 ToRGB), `EqualLinear(style)` and —
for the 13 demodulated ones — `rsqrt(sum (scale * W * s)^2 + eps)` (reference layers.py:293-300).  Each is a
[B, 512] x [512, C] product: 8 MFLOP at batch 16, i.e. pure launch latency and a 1 MB weight read at ~100 GB/s when it
is launched on its own.  One generator forward + backward spent ~120 such launches (~1 ms of 33 at batch 16), and the
path-length iteration, whose backward is itself recorded and differentiated, ~1 500 more (every tiny tensor operation
of the re-derived vector-Jacobian products, per layer).

Neither depends on the activations, only on the latent and the weights, so `build()` evaluates them for ALL layers
right after the mapping network.  Round 6 (`_build_native`, op/bankmm.py + csrc/bank_mm.hip):

    modulation   ONE launch: every layer's s_l = scale * latent[:, idx_l] @ M_l^T + lr_mul * b_l, written as blocks of one
                 flat buffer; the weights are read where they lie (a table of per-layer pointers in the kernel argument)
    demodulation s^2 on the flat buffer (one element-wise launch), ONE launch for q_l = s_l^2 @ Wsq_l of all demodulated
                 layers, rsqrt(q + eps) on the flat result; per-layer d_l are views
    weight prep  unchanged: one k_wprep launch per layer (or the weight bank's batched one) gives wt and Wsq

    gradients    the two product families are closed under differentiation (bankmm.ModNT/NN/TN, DemNN/NT/TN), the
                 element-wise steps are plain tensor algebra on flat buffers: any order of gradient costs a handful of
                 launches per pass, the layers that share a latent row add their gradients inside one kernel in a fixed
                 order, and no vendor GEMM is called (round 3-5: stacked weights + rocBLAS strided-batched baddbmm / bmm,
                 ~25 stack / cat copies per forward — `_build_stacked`, kept behind SR_STYLE_BANK=stacked for A/B runs
                 and for shapes outside the kernels' alignment rules).

Each layer then receives a `StylePack` in place of its latent row; `ModulatedConv2d` uses the precomputed values when
it is handed one.  SR_STYLE_BANK=0 restores the per-layer evaluation.  CPU tensors never take this path (reference
formulation).
"""
import os

import torch

from . import bankmm as _bank
from . import style as _style


class _Unbind0(torch.autograd.Function):
    """x.unbind(0) whose backward is `_Stack0` (and vice versa).  torch's own stack / unbind differentiate through
    narrow + squeeze views, whose SECOND derivative materialises a zero-filled copy of the stacked tensor per layer
    (select_backward: zeros + copy + accumulate, ~100 launches in the path-length sweep)."""

    @staticmethod
    def forward(ctx, x):
        return x.unbind(0)

    @staticmethod
    def backward(ctx, *grads):
        return _Stack0.apply(*grads)


class _Stack0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ts):
        return torch.stack(ts)

    @staticmethod
    def backward(ctx, g):
        # one re-layout of the stacked gradient when it arrives strided (the modulation weights are consumed transposed:
        # their gradient is a transposed view, and AccumulateGrad would otherwise clone every layer's slice to meet the
        # parameter's layout — 15 copies of 1 MB per backward at 256^2)
        return _Unbind0.apply(g.contiguous())


def stack0(ts):
    return _Stack0.apply(*ts)


def unbind0(x):
    return _Unbind0.apply(x)


class StylePack:
    """Precomputed per-layer style data: s [B, Cin] (modulation), d [B, Cout] | None (demodulation), wt / wsq from
    weight_prep (None for the ToRGB 1x1 layers, which prepare their three rows themselves)."""

    __slots__ = ("s", "d", "wt", "wsq")

    def __init__(self, s, d=None, wt=None, wsq=None):
        self.s, self.d, self.wt, self.wsq = s, d, wt, wsq


def enabled():
    return os.environ.get("SR_STYLE_BANK", "1") != "0"


def build(layers, latent):
    """layers: [(ModulatedConv2d, latent index)] in call order; latent [B, n_latent, D] (device tensor).
    Returns one StylePack per entry."""
    if os.environ.get("SR_STYLE_BANK", "1") != "stacked":
        packs = _build_native(layers, latent)
        if packs is not None:
            return packs
    return _build_stacked(layers, latent)


def _is_torgb(m):
    return m.kernel_size == 1 and not m.demodulate and not m.upsample and not m.downsample


def _build_native(layers, latent):
    """The table-driven kernels of csrc/bank_mm.hip; None when a shape falls outside their rules (the stacked form runs)."""
    mods = [m.modulation for m, _ in layers]
    if not mods:
        return []
    scale, lr_mul, has_bias = float(mods[0].scale), float(mods[0].lr_mul), mods[0].bias is not None
    if any(float(md.scale) != scale or float(md.lr_mul) != lr_mul or (md.bias is not None) != has_bias
           or md.activation for md in mods):
        return None
    weights = [md.weight for md in mods]
    biases = [md.bias for md in mods] if has_bias else None
    if latent.dim() != 3 or not _bank.modulation_supported(latent, weights, biases):
        return None
    batch = latent.shape[0]
    s_flat, s_of = _bank.modulation(latent, [li for _, li in layers], weights, biases, scale, lr_mul)
    n = len(layers)
    packs = [None] * n
    dem = []
    for i, (m, _) in enumerate(layers):
        if _is_torgb(m):
            packs[i] = StylePack(s_of[i])                                          # ToRGB: modulation only
            continue
        wt, wsq = m._wprep(m.demodulate)
        packs[i] = StylePack(s_of[i], None, wt, wsq)
        if m.demodulate:
            dem.append(i)
    if dem:
        eps = float(layers[dem[0]][0].eps)
        mats = [packs[i].wsq for i in dem]
        if any(float(layers[i][0].eps) != eps for i in dem) or not _bank.demod_supported(mats):
            for i in dem:                                                          # per-layer fused kernel (op.style)
                p = packs[i]
                p.d = (_style.demod_scale(p.s, p.wsq, float(layers[i][0].eps)) if _style.demod_supported(p.s, p.wsq)
                       else torch.rsqrt(torch.matmul(p.s * p.s, p.wsq) + float(layers[i][0].eps)))
            return packs
        offs, o = [], 0
        for m, _ in layers:
            offs.append(o)
            o += batch * m.in_channel
        q_flat, _ = _bank.demod_products(s_flat * s_flat, batch, [offs[i] for i in dem], mats)
        d_all = torch.rsqrt(q_flat + eps)
        for i, d in zip(dem, _bank._blocks(d_all, batch, [int(mt.shape[1]) for mt in mats])):
            packs[i].d = d
    return packs


def _build_stacked(layers, latent):
    """Round 3-5: stacked weights, one strided-batched library GEMM per group of equally shaped layers."""
    rows = latent.unbind(1)                       # one unbind (its backward is one stack)
    n = len(layers)
    s_of = [None] * n
    groups = {}
    for i, (m, _) in enumerate(layers):
        mod = m.modulation
        groups.setdefault((m.in_channel, float(mod.scale), float(mod.lr_mul), mod.bias is not None), []).append(i)
    for (_, scale, lr_mul, has_bias), idxs in groups.items():
        mods = [layers[i][0].modulation for i in idxs]
        lat = stack0([rows[layers[i][1]] for i in idxs])                           # [L, B, D]
        w = stack0([m.weight for m in mods]).transpose(1, 2)                       # [L, D, C]
        if has_bias:
            bias = stack0([m.bias for m in mods]).unsqueeze(1)                     # [L, 1, C]
            if lr_mul != 1.0:
                bias = bias * lr_mul
            s = torch.baddbmm(bias, lat, w, alpha=scale)                           # [L, B, C]
        else:
            s = torch.bmm(lat, w) * scale
        for i, row in zip(idxs, unbind0(s)):        # one unbind (backward: one stack), not a select per layer
            s_of[i] = row
    packs = [None] * n
    dgroups = {}
    for i, (m, _) in enumerate(layers):
        if m.kernel_size == 1 and not m.demodulate and not m.upsample and not m.downsample:
            packs[i] = StylePack(s_of[i])                                          # ToRGB: modulation only
            continue
        wt, wsq = m._wprep(m.demodulate)
        packs[i] = StylePack(s_of[i], None, wt, wsq)
        if m.demodulate:
            dgroups.setdefault((m.in_channel, m.out_channel, float(m.eps)), []).append(i)
    for (_, _, eps), idxs in dgroups.items():
        if len(idxs) == 1:
            p = packs[idxs[0]]
            p.d = (_style.demod_scale(p.s, p.wsq, eps) if _style.demod_supported(p.s, p.wsq)
                   else torch.rsqrt(torch.matmul(p.s * p.s, p.wsq) + eps))
            continue
        ss = stack0([packs[i].s for i in idxs])                                    # [L, B, Ci]
        wq = stack0([packs[i].wsq for i in idxs])                                  # [L, Ci, Co]
        d = torch.rsqrt(torch.bmm(ss * ss, wq) + eps)                              # [L, B, Co]
        for i, row in zip(idxs, unbind0(d)):
            packs[i].d = row
    return packs
