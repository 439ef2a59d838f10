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
