"""All modulation linears and demodulation factors of a generator pass as a handful of batched GEMMs.

The reference evaluates, per modulated convolution (26 at 256^2: 13 StyledConv + 13 ToRGB), `EqualLinear(style)` and —
for the 13 demodulated ones — `rsqrt(sum (scale * W * s)^2 + eps)` (reference layers.py:293-300).  Each is a
[B, 512] x [512, C] product: 8 MFLOP at batch 16, i.e. pure launch latency and a 1 MB weight read at ~100 GB/s when it
is launched on its own.  One generator forward + backward spent ~120 such launches (~1 ms of 33 at batch 16), and the
path-length iteration, whose backward is itself recorded and differentiated, ~1 500 more (every tiny tensor operation
of the re-derived vector-Jacobian products, per layer).

Neither depends on the activations, only on the latent and the weights, so `build()` evaluates them for ALL layers
right after the mapping network:

    modulation   grouped by input channels (512 / 256 / 128): s[l] = scale * latent[:, idx_l] @ M_l^T + b_l  as ONE
                 `baddbmm` per group over stacked weights (plain library GEMM: rocBLAS / hipBLASLt strided-batched)
    demodulation grouped by (Cin, Cout): d[l] = rsqrt((s_l * s_l) @ Wsq_l + eps) as one `bmm` per group of >= 2 layers
                 (single layers keep the fused kernel of op.style)
    weight prep  unchanged: one k_wprep launch per layer gives the tap-major weights and Wsq (csrc/weight_prep.hip)

Everything is ordinary differentiable tensor algebra, so gradients of any order are batched the same way (the
second-order graph of the path-length regulariser shrinks by the number of layers in a group).  Each layer then
receives a `StylePack` in place of its latent row; `ModulatedConv2d` uses the precomputed values when it is handed
one.  SR_STYLE_BANK=0 restores the per-layer evaluation.  CPU tensors never take this path (reference formulation).
"""
import os

import torch

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
