"""CPU: op._dispatch.DerivedCache — LRU eviction that never frees an entry a captured hipGraph may reference."""
import torch

from stylerenderer_amd.op import _dispatch
import importlib

ufd = importlib.import_module("stylerenderer_amd.op.upfirdn2d")      # (op.upfirdn2d is the function)


def test_lru_eviction_keeps_recent_and_pinned(monkeypatch):
    c = _dispatch.DerivedCache(3)
    for k in range(3):
        c.put(k, ("v%d" % k,))
    assert c.get(0) == ("v0",)                   # 0 becomes most recent
    c.put(3, ("v3",))
    assert c.get(1) is None and c.get(0) is not None and len(c) == 3
    # entries touched during stream capture are pinned: later churn cannot evict them
    monkeypatch.setattr(_dispatch, "_capturing", lambda: True)
    assert c.get(2) == ("v2",)
    c.put(10, ("captured",))
    monkeypatch.setattr(_dispatch, "_capturing", lambda: False)
    for k in range(20, 40):
        c.put(k, (k,))
    assert c.get(2) == ("v2",) and c.get(10) == ("captured",)
    assert len(c) == 3 + 2                       # capacity counts unpinned entries only
    c.clear()
    assert len(c) == 2


def test_flipped_taps_cache_hits_by_address_and_version():
    k = torch.arange(16.0).reshape(4, 4)
    a = ufd.flipped(k)
    assert torch.equal(a, torch.flip(k, [0, 1])) and ufd.flipped(k) is a
    k.mul_(2)                                    # in-place change bumps the version: no stale taps
    b = ufd.flipped(k)
    assert b is not a and torch.equal(b, torch.flip(k, [0, 1]))
    many = [torch.full((4, 4), float(i)) for i in range(200)]
    for t in many:
        ufd.flipped(t)
    assert len(ufd._FLIP_CACHE) <= 64 + len(ufd._FLIP_CACHE.pinned) + 1
