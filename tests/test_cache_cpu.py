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


def test_pins_are_released_with_the_graph_that_took_them(monkeypatch):
    """pins used to live for the life of the process.  A capture made through graphs.capture opens a
    PinScope; when that graph dies its entries become evictable again, while entries shared with a graph that is still
    alive stay — and a scope keeps the VALUE it read alive even if the cache replaced the entry meanwhile."""
    c = _dispatch.DerivedCache(2)
    c.put("shared", ("s",))
    a, b = _dispatch.PinScope(), _dispatch.PinScope()
    monkeypatch.setattr(_dispatch, "_capturing", lambda: True)
    with _dispatch.pin_scope(a):
        c.get("shared")
        c.get("shared")                                  # pinned once per scope
        c.put("only_a", ("a",))
        held = _dispatch.hold_for_capture(("frozen weights",))
    with _dispatch.pin_scope(b):
        c.get("shared")
    monkeypatch.setattr(_dispatch, "_capturing", lambda: False)
    assert c.pinned == {"shared": 2, "only_a": 1} and a.keep == [held]
    for k in range(10):
        c.put(k, (k,))
    assert c.get("shared") and c.get("only_a") and len(c) == 2 + 2
    c.put("only_a", ("replaced",))                       # the live graph a still owns the old value
    assert ("a",) in [v for _c, _k, v in a.items]
    a.release()
    a.release()                                          # idempotent (weakref.finalize + explicit)
    assert c.pinned == {"shared": 1} and a.keep == [] and len(c) == 2 + 1
    b.release()
    assert c.pinned == {}
    c.put("x", (0,))
    assert len(c) == 2
