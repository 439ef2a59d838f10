"""Shared helpers of the operator wrappers."""
import collections
import contextlib

import torch

from .. import _lib


def is_device_tensor(t):
    return t.device.type == "cuda"


def require_f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError("%s: expected a float32 tensor, got %s" % (name, t.dtype))


@contextlib.contextmanager
def on_device_of(t):
    """Makes the tensor's GPU current for the duration of a native launch (one process per GPU
    normally makes this a no-op)."""
    idx = t.device.index
    if idx is None or idx == torch.cuda.current_device():
        yield
    else:
        with torch.cuda.device(idx):
            yield


def stream_of(t):
    return _lib.current_stream(t.device)


def strict_native():
    """SR_STRICT_NATIVE=1 (set by tests/conftest.py for the GPU suite and by bench.py): a device tensor that would
    leave this library's kernels for a MIOpen / rocBLAS fallback raises instead — the fallbacks exist for shapes the
    generator / discriminator never produce, and a benchmark or parity run must not take one silently."""
    import os

    return os.environ.get("SR_STRICT_NATIVE", "0") == "1"


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class PinScope:
    """What one captured graph keeps alive: the cache entries it read or created (their addresses are baked into the
    hipGraph) and any other tensor handed to `hold_for_capture`.  `graphs.capture` opens one around the capture and
    releases it when the graph object dies (weakref.finalize) — a LatentInverter per image or a re-capture after
    load_checkpoint then gives its pins back instead of growing device memory for the life of the process."""

    def __init__(self):
        self.items = []          # (cache, key, value): value held strongly — survives the cache replacing the entry
        self.keep = []
        self.released = False

    def release(self):
        if self.released:
            return
        self.released = True
        for cache, key, _value in self.items:
            cache._unpin(key)
        self.items, self.keep = [], []


_SCOPES = []


@contextlib.contextmanager
def pin_scope(scope):
    _SCOPES.append(scope)
    try:
        yield scope
    finally:
        _SCOPES.pop()


def hold_for_capture(obj):
    """Keeps `obj` (tensors a graph under capture will read at replay) alive as long as that graph."""
    if _SCOPES and _capturing():
        _SCOPES[-1].keep.append(obj)
    return obj


class DerivedCache:
    """LRU cache of tensors derived from long-lived inputs (flipped FIR taps, per-topology incidence lists), keyed by
    the source tensor's address / version; the value must hold the source so that its address cannot be reused.

    Entries that are read or created while the current stream is being CAPTURED are pinned for as long as the graph
    under capture lives (`PinScope`; for the life of the process when the capture did not go through
    `graphs.capture`): the hipGraph holds their addresses, so evicting (freeing) them would make every later replay
    read recycled memory.  (Round 2 cleared the whole dict at a size threshold: a later test's modules pushed a live
    graph's flipped taps out and the replays went NaN.)  Only unpinned entries count against `capacity`."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.data = collections.OrderedDict()
        self.pinned = {}                         # key -> number of live graphs (or 1 << 30: no scope, for ever)

    def _pin(self, key, value):
        if not _capturing():
            return
        if not _SCOPES:
            self.pinned[key] = 1 << 30
            return
        scope = _SCOPES[-1]
        if not any(c is self and k == key for c, k, _ in scope.items):
            scope.items.append((self, key, value))
            self.pinned[key] = self.pinned.get(key, 0) + 1

    def _unpin(self, key):
        n = self.pinned.get(key, 0) - 1
        if n > 0:
            self.pinned[key] = n
        else:
            self.pinned.pop(key, None)
            self._evict()

    def _evict(self):
        excess = len(self.data) - sum(1 for k in self.data if k in self.pinned) - self.capacity
        if excess > 0:
            for k in [k for k in self.data if k not in self.pinned][:excess]:
                del self.data[k]

    def get(self, key):
        hit = self.data.get(key)
        if hit is not None:
            self.data.move_to_end(key)
            self._pin(key, hit)
        return hit

    def put(self, key, value):
        self.data[key] = value
        self._pin(key, value)
        self._evict()
        return value

    def __len__(self):
        return len(self.data)

    def clear(self):
        """Drops the unpinned entries (tests)."""
        for k in [k for k in self.data if k not in self.pinned]:
            del self.data[k]


# ---- gradients nobody asked for ----------------------------------------------------------------------------------------
# `ctx.needs_input_grad` of a Python autograd Function is fixed at FORWARD time (does the input require grad at all),
# while the engine knows per backward pass which edges it will follow: `autograd.grad(image, latents, create_graph=True)`
# of the path-length regulariser (reference train.py:118-134) and `autograd.grad(real_pred.sum(), real_img,
# create_graph=True)` of R1 (train.py:110-116) never use a weight / bias / noise-strength gradient, yet every layer node
# would compute (and record) its weight-gradient kernels in that first pass.  `wanted(ctx)` is `needs_input_grad`
# intersected with "the engine will execute the node this gradient flows to in the pass that is running now".
def mark_inputs(ctx, *args):
    """Call first thing in forward(ctx, *args): remembers which arguments are tensors — `ctx.next_functions` has one
    edge per TENSOR argument, `ctx.needs_input_grad` one flag per argument."""
    ctx._sr_tmask = tuple(isinstance(a, torch.Tensor) for a in args)


def _prune_enabled():
    import os

    return os.environ.get("SR_PRUNE_GRADS", "1") != "0"


def wanted(ctx):
    """`ctx.needs_input_grad`, with False also for inputs whose gradient the running backward pass would discard.  Safe
    by construction: anything the engine cannot answer (no mask recorded, edge count mismatch, a leaf that is itself one
    of the `inputs=` of autograd.grad — the engine refuses that query) counts as needed."""
    needs = tuple(ctx.needs_input_grad)
    mask = getattr(ctx, "_sr_tmask", None)
    if mask is None or not _prune_enabled():
        return needs
    try:
        edges = ctx.next_functions
    except Exception:                                   # not a node (a stand-in context): nothing to prune
        return needs
    if len(edges) != sum(mask):
        return needs
    out, e = [], 0
    for i, nd in enumerate(needs):
        is_t = mask[i] if i < len(mask) else False
        if nd and is_t:
            fn = edges[e][0]
            if fn is None:
                nd = False
            else:
                try:
                    nd = bool(torch._C._will_engine_execute_node(fn))
                except RuntimeError:
                    nd = True
        if is_t:
            e += 1
        out.append(bool(nd))
    return tuple(out)
