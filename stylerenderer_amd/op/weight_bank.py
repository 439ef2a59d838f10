"""Every convolution weight of a network prepared by one or two launches per forward pass.

Per convolution and pass the MFMA path needs the tap-major scaled weights (+ the demodulation matrix) — `k_wprep` — and,
for the data gradient, their channel-transposed (tap-reversed) adjoint — `k_wadjoint` (op/weight_prep.py).  Each is a
~5 us launch on a few hundred KB, and a training iteration at the reference's per-GPU batch (4 images: BASELINE
config[2]) issues ~230 of them in its forward passes: launch count, not bytes.  Neither depends on activations, only on
the parameters, so a network's forward opens a SCOPE that prepares all of its convolutions up front:

    first pass of a network   runs layer by layer and records which modules asked for prepared weights
    every later pass          ONE batched `k_wprep` launch for all recorded layers and, when gradients are recorded, ONE
                              batched `k_wadjoint` launch (C ABI sr_weight_prep_batch / sr_weight_adjoint_batch:
                              per-layer pointers by value in the kernel argument; same arithmetic per element, so the
                              results are bit-identical to the per-layer launches)

The batched launches are plain data producers (no autograd node).  A layer that picks its entry up gets it through
`_WPrepUse`, a per-layer node created AT THE POINT OF USE whose forward returns the prepared tensors and whose backward is
the layer's own `k_wprep_bwd` pull-back — exactly the node `weight_prep` would have recorded.  (A batched backward node
was measured and dropped: autograd runs ready nodes in reverse creation order, a node created at the start of the
forward therefore runs at the very END of the backward, and every convolution weight gradient — most of the gradient
bytes — reached the parameters only then: the bucketed all-reduce that overlaps the backward,
distributed.BucketedGradReducer, had nothing to send until the replay was over.)

The banked adjoint is data too: `ConvFn.backward` uses it when the backward is NOT being recorded, and re-derives the
adjoint differentiably from `wt` when it is (R1, path-length regulariser: the gradient with respect to the weights
flows through the data-gradient convolution's weights).

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


def _arr(ctype, vals):
    return (ctype * len(vals))(*vals)


def _ptrs(ts):
    return _arr(ctypes.c_void_p, [t.data_ptr() if t is not None and t.numel() else None for t in ts])


def _pitch(n):
    return (n + 3) // 4 * 4


def prep_batch(weights, scales, want_sqs):
    """[(wt [k*k, Ci, Co] view of the padded buffer, wsq [Ci, Co] | None)] for n weights [Co, Ci, k, k]: one launch."""
    dims = [_wp._as3(w) for w in weights]
    ws = [w.detach().contiguous() for w in weights]
    lds = [_pitch(co) for co, _, _ in dims]
    sizes = [(k * k * ci * ld, _pitch(ci * co) if sq else 0) for (co, ci, k), ld, sq in zip(dims, lds, want_sqs)]
    store = torch.empty(sum(a + b for a, b in sizes), dtype=ws[0].dtype, device=ws[0].device)
    wts, wsqs, outs, off = [], [], [], 0
    for (co, ci, k), ld, (a, b), sq in zip(dims, lds, sizes, want_sqs):
        wt = store[off:off + a].view(k * k, ci, ld)
        wsq = store[off + a:off + a + ci * co].view(ci, co) if sq else None
        off += a + b
        wts.append(wt)
        wsqs.append(wsq)
        outs.append((wt if ld == co else wt[:, :, :co], wsq))
    with on_device_of(store):
        rc = _lib.lib().sr_weight_prep_batch(
            len(ws), _ptrs(wts), _ptrs(wsqs), _ptrs(ws), _arr(ctypes.c_float, [float(s) for s in scales]),
            _arr(ctypes.c_int64, [d[0] for d in dims]), _arr(ctypes.c_int64, [d[1] for d in dims]),
            _arr(ctypes.c_int, [d[2] for d in dims]), _arr(ctypes.c_int64, lds), stream_of(store))
    _lib.check(rc, "sr_weight_prep_batch")
    return outs


def adjoint_batch(wts, flips):
    """[adjoint_i [taps, N, C]] of wt_i [taps, C, N] (taps reversed where flips[i]): one launch."""
    srcs = []
    for wt in wts:
        taps, c, n = wt.shape
        ok = wt.stride(2) == 1 and wt.stride(0) == c * wt.stride(1) and wt.stride(1) >= n
        srcs.append(wt.detach() if ok else wt.detach().contiguous())
    shapes = [tuple(wt.shape) for wt in wts]
    ldcs = [_pitch(c) for _, c, _ in shapes]
    sizes = [t * n * ldc for (t, _, n), ldc in zip(shapes, ldcs)]
    store = torch.empty(sum(sizes), dtype=srcs[0].dtype, device=srcs[0].device)
    full, outs, off = [], [], 0
    for (t, c, n), ldc, m in zip(shapes, ldcs, sizes):
        a = store[off:off + m].view(t, n, ldc)
        off += m
        full.append(a)
        outs.append(a if ldc == c else a[:, :, :c])
    with on_device_of(store):
        rc = _lib.lib().sr_weight_adjoint_batch(
            len(srcs), _ptrs(full), _ptrs(srcs), _arr(ctypes.c_int64, [s[0] for s in shapes]),
            _arr(ctypes.c_int64, [s[1] for s in shapes]), _arr(ctypes.c_int64, [s[2] for s in shapes]),
            _arr(ctypes.c_int64, [s.stride(1) for s in srcs]), _arr(ctypes.c_int64, ldcs),
            _arr(ctypes.c_int, [int(bool(f)) for f in flips]), stream_of(store))
    _lib.check(rc, "sr_weight_adjoint_batch")
    return outs


class _WPrepUse(Function):
    """(weight, prepared wt, prepared wsq) -> (wt, wsq): the autograd node of weight_prep._WPrep around tensors that
    the scope's batched launch has already filled."""

    @staticmethod
    def forward(ctx, weight, wt, wsq, scale, want_sq):
        ctx.save_for_backward(weight)
        ctx.set_materialize_grads(False)
        ctx.scale, ctx.want_sq = float(scale), bool(want_sq)
        out_sq = wsq.detach() if want_sq else weight.new_empty(0)
        if not want_sq:
            ctx.mark_non_differentiable(out_sq)
        return wt.detach(), out_sq

    @staticmethod
    def backward(ctx, gwt, gwsq):
        (weight,) = ctx.saved_tensors
        if not ctx.want_sq:
            gwsq = None
        if gwt is None and gwsq is None:
            return None, None, None, None, None
        return _wp._WPrepBwd.apply(gwt, gwsq, weight, ctx.scale), None, None, None, None


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
        outs = prep_batch([m.weight for m, _ in live], [m.scale for m, _ in live], [sq for _, sq in live])
        adjs = [None] * len(live)
        if torch.is_grad_enabled():
            flips = [_flip_of(m) for m, _ in live]
            adjs = list(zip(flips, adjoint_batch([wt for wt, _ in outs], flips)))
        for (m, _), (wt, wsq), adj in zip(live, outs, adjs):
            m._bank = (wt, wsq, adj)
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
    """The scope's entry for `module` as (wt, wsq) behind the layer's own autograd node, the banked adjoint attached
    to wt; None when the scope holds nothing usable for this request."""
    hit = getattr(module, "_bank", None)
    if hit is None or (want_sq and hit[1] is None):
        return None
    wt, wsq, adj = hit
    weight = module.weight
    if weight.requires_grad and torch.is_grad_enabled():
        wt, wsq = _WPrepUse.apply(weight, wt, wsq, module.scale, want_sq)
    if adj is not None:
        wt._sr_adj = adj
    return wt, (wsq if want_sq else None)
