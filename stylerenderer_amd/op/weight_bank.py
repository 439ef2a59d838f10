"""Every convolution weight of a network prepared by a handful of launches per pass.

Per convolution and pass the MFMA path needs the tap-major scaled weights (+ the demodulation matrix) — `k_wprep` —, for
the data gradient their channel-transposed (tap-reversed) adjoint — `k_wadjoint` — and, in the backward, the pull-back
of both cotangents onto the parameter — `k_wprep_bwd` (op/weight_prep.py).  Each is a ~5 us launch on a few hundred KB,
and a training iteration at the reference's per-GPU batch (4 images: BASELINE config[2]) issues ~290 of them: launch
count, not bytes.  None depends on activations, only on the parameters, so a network's forward opens a SCOPE that
prepares all of its convolutions up front:

    first pass of a network   runs layer by layer and records which modules asked for prepared weights, in call order
    every later pass          SR_WEIGHT_BANK_GROUPS (4) groups of consecutive layers; per group ONE batched `k_wprep`
                              launch, ONE batched `k_wadjoint` launch when gradients are recorded, and in the backward
                              ONE batched `k_wprep_bwd` launch (C ABI sr_weight_prep_batch / _adjoint_batch / _bwd_batch:
                              per-layer pointers by value in the kernel argument; same arithmetic per element, so the
                              results are bit-identical to the per-layer launches)

Groups, not one launch for the whole network: a group's backward node runs when the LAST of its layers has produced
its weight gradient.  With a single node every convolution weight of the network — most of the gradient bytes — would
reach the parameters only at the very end of the backward, and the bucketed all-reduce that overlaps the backward
(distributed.BucketedGradReducer) would have nothing to send until then.  Groups of consecutive layers keep the
arrival order of the buckets.

Layers pick their entry up through `weight_prep.weight_prep_cached`; `ConvFn` finds the adjoint on the prepared tensor.
Modules frozen with `freeze_prepared_weights` keep their version-keyed cache.  SR_WEIGHT_BANK=0 disables the scope.
CPU tensors never take this path.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import on_device_of, stream_of
from . import weight_prep as _wp

_REC = None            # recording list of the scope that is open on a network without a plan


def enabled():
    return os.environ.get("SR_WEIGHT_BANK", "1") != "0"


def n_groups():
    return max(1, int(os.environ.get("SR_WEIGHT_BANK_GROUPS", "4")))


def _arr(ctype, vals):
    return (ctype * len(vals))(*vals)


def _ptrs(ts):
    return _arr(ctypes.c_void_p, [t.data_ptr() if t is not None and t.numel() else None for t in ts])


def _pitch(n):
    return (n + 3) // 4 * 4


class _WPrepBatch(Function):
    """weights [Co, Ci, k, k] x n -> (wt_0, wsq_0, wt_1, wsq_1, ...) carved from one buffer."""

    @staticmethod
    def forward(ctx, meta, *weights):
        dims = [_wp._as3(w) for w in weights]
        ws = [w.contiguous() for w in weights]
        lds = [_pitch(co) for co, _, _ in dims]
        sizes = []
        for (co, ci, k), ld, (_, want_sq) in zip(dims, lds, meta):
            sizes.append((k * k * ci * ld, _pitch(ci * co) if want_sq else 0))
        store = torch.empty(sum(a + b for a, b in sizes), dtype=ws[0].dtype, device=ws[0].device)
        wts, wsqs, outs, off = [], [], [], 0
        for (co, ci, k), ld, (a, b), (_, want_sq) in zip(dims, lds, sizes, meta):
            wt = store[off:off + a].view(k * k, ci, ld)
            wsq = store[off + a:off + a + ci * co].view(ci, co) if want_sq else store.new_empty(0)
            off += a + b
            wts.append(wt)
            wsqs.append(wsq)
            outs += [wt if ld == co else wt[:, :, :co], wsq]
        with on_device_of(store):
            rc = _lib.lib().sr_weight_prep_batch(
                len(ws), _ptrs(wts), _ptrs(wsqs), _ptrs(ws), _arr(ctypes.c_float, [float(s) for s, _ in meta]),
                _arr(ctypes.c_int64, [d[0] for d in dims]), _arr(ctypes.c_int64, [d[1] for d in dims]),
                _arr(ctypes.c_int, [d[2] for d in dims]), _arr(ctypes.c_int64, lds), stream_of(store))
        _lib.check(rc, "sr_weight_prep_batch")
        ctx.save_for_backward(*weights)
        ctx.meta = meta
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*[q for q, (_, want_sq) in zip(wsqs, meta) if not want_sq])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        weights = ctx.saved_tensors
        n = len(weights)
        gwts = [grads[2 * i] for i in range(n)]
        gwsqs = [grads[2 * i + 1] if ctx.meta[i][1] else None for i in range(n)]
        live = [i for i in range(n) if ctx.needs_input_grad[1 + i] and (gwts[i] is not None or gwsqs[i] is not None)]
        out = [None] * n
        if not live:
            return (None,) + tuple(out)
        if torch.is_grad_enabled():
            # the backward itself is being recorded (path-length regulariser): per-layer differentiable nodes
            for i in live:
                out[i] = _wp._WPrepBwd.apply(gwts[i], gwsqs[i], weights[i], ctx.meta[i][0])
            return (None,) + tuple(out)
        dims = [_wp._as3(weights[i]) for i in live]
        ws = [weights[i].contiguous() for i in live]
        gts = [gwts[i].contiguous() if gwts[i] is not None else None for i in live]
        gqs = [gwsqs[i].contiguous() if gwsqs[i] is not None else None for i in live]
        numels = [_pitch(w.numel()) for w in ws]
        store = torch.empty(sum(numels), dtype=ws[0].dtype, device=ws[0].device)
        gws, off = [], 0
        for w, m in zip(ws, numels):
            gws.append(store[off:off + w.numel()].view(w.shape))
            off += m
        with on_device_of(store):
            rc = _lib.lib().sr_weight_prep_bwd_batch(
                len(ws), _ptrs(gws), _ptrs(gts), _ptrs(gqs), _ptrs(ws),
                _arr(ctypes.c_float, [float(ctx.meta[i][0]) for i in live]),
                _arr(ctypes.c_int64, [d[0] for d in dims]), _arr(ctypes.c_int64, [d[1] for d in dims]),
                _arr(ctypes.c_int, [d[2] for d in dims]), _arr(ctypes.c_int64, [d[0] for d in dims]),
                stream_of(store))
        _lib.check(rc, "sr_weight_prep_bwd_batch")
        for i, g in zip(live, gws):
            out[i] = g.view(weights[i].shape)
        return (None,) + tuple(out)


class _AdjointBatch(Function):
    """wt_i [taps, C, N] -> adjoint_i [taps, N, C] (taps reversed where flips[i]) for n layers in one launch."""

    @staticmethod
    def forward(ctx, flips, *wts):
        srcs = []
        for wt in wts:
            taps, c, n = wt.shape
            ok = wt.stride(2) == 1 and wt.stride(0) == c * wt.stride(1) and wt.stride(1) >= n
            srcs.append(wt if ok else wt.contiguous())
        shapes = [tuple(wt.shape) for wt in wts]
        ldcs = [_pitch(c) for _, c, _ in shapes]
        sizes = [t * n * ldc for (t, _, n), ldc in zip(shapes, ldcs)]
        store = torch.empty(sum(sizes), dtype=wts[0].dtype, device=wts[0].device)
        full, outs, off = [], [], 0
        for (t, c, n), ldc, m in zip(shapes, ldcs, sizes):
            a = store[off:off + m].view(t, n, ldc)
            off += m
            full.append(a)
            outs.append(a if ldc == c else a[:, :, :c])
        with on_device_of(store):
            rc = _lib.lib().sr_weight_adjoint_batch(
                len(wts), _ptrs(full), _ptrs(srcs), _arr(ctypes.c_int64, [s[0] for s in shapes]),
                _arr(ctypes.c_int64, [s[1] for s in shapes]), _arr(ctypes.c_int64, [s[2] for s in shapes]),
                _arr(ctypes.c_int64, [s.stride(1) for s in srcs]), _arr(ctypes.c_int64, ldcs),
                _arr(ctypes.c_int, [int(bool(f)) for f in flips]), stream_of(store))
        _lib.check(rc, "sr_weight_adjoint_batch")
        ctx.flips = flips
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        # reached only when a data-gradient convolution is itself differentiated with respect to its weights
        # (second-order passes): the permutation is its own inverse
        return (None,) + tuple(None if g is None else _wp.adjoint(g, f) for g, f in zip(grads, ctx.flips))


def _flip_of(module):
    """Tap reversal of the module's data-gradient weights: only the stride-1 3x3 correlation flips (op.conv)."""
    if hasattr(module, "_geom"):
        return module._geom() == "c3"
    return (getattr(module, "kernel_size", 1) == 3 and not getattr(module, "upsample", False)
            and not getattr(module, "downsample", False))


def note(module, want_sq):
    """Called by weight_prep_cached on a miss: remembers the request while a recording scope is open."""
    if _REC is not None:
        _REC.append((module, bool(want_sq)))


class Scope:
    """`with weight_bank.Scope(net): ...` around a network's forward."""

    def __init__(self, net):
        self.net = net
        self.recording = False
        self.filled = []

    def __enter__(self):
        global _REC
        net = self.net
        if not enabled() or _REC is not None:
            return self
        plan = getattr(net, "_bank_plan", None)
        if plan is None:
            _REC = []
            self.recording = True
            return self
        live = [(m, sq) for m, sq in plan
                if not getattr(m, "_frozen_weights", False) and m.weight.device.type == "cuda"
                and m.weight.dtype == torch.float32]
        if not live:
            return self
        want_adj = torch.is_grad_enabled()
        g = min(n_groups(), len(live))
        total = sum(m.weight.numel() for m, _ in live)
        groups, cur, acc = [], [], 0
        for m, sq in live:                                   # consecutive layers, ~equal bytes per group
            cur.append((m, sq))
            acc += m.weight.numel()
            if acc * g >= total * (len(groups) + 1) and len(groups) < g - 1:
                groups.append(cur)
                cur = []
        if cur:
            groups.append(cur)
        for grp in groups:
            meta = tuple((float(m.scale), sq) for m, sq in grp)
            outs = _WPrepBatch.apply(meta, *[m.weight for m, _ in grp])
            wts = [outs[2 * i] for i in range(len(grp))]
            adjs = [None] * len(grp)
            if want_adj:
                adjs = _AdjointBatch.apply(tuple(_flip_of(m) for m, _ in grp), *wts)
            for i, (m, sq) in enumerate(grp):
                m._bank = (wts[i], outs[2 * i + 1] if sq else None, None if adjs[i] is None else (_flip_of(m), adjs[i]))
                self.filled.append(m)
        return self

    def __exit__(self, *exc):
        global _REC
        if self.recording:
            seen, plan = {}, []
            for m, sq in _REC:
                if id(m) in seen:
                    plan[seen[id(m)]] = (m, plan[seen[id(m)]][1] or sq)
                else:
                    seen[id(m)] = len(plan)
                    plan.append((m, sq))
            _REC = None
            if exc[0] is None and plan:          # (a CPU pass records nothing: the next device pass records again)
                self.net._bank_plan = plan
        for m in self.filled:
            m._bank = None
        self.filled = []
        return False


def lookup(module, want_sq):
    """The scope's entry for `module`: (wt, wsq) with the adjoint attached to wt, or None."""
    hit = getattr(module, "_bank", None)
    if hit is None or (want_sq and hit[1] is None):
        return None
    wt, wsq, adj = hit
    if adj is not None:
        wt._sr_adj = adj
    return wt, (wsq if want_sq else None)
