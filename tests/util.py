"""Helpers shared by the tests."""
import numpy as np


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def max_ulp(a, b):
    """Largest distance in float32 units-in-the-last-place between two arrays."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)

    def key(x):
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, np.int64(-(2 ** 31)) - i, i)

    return int(np.abs(key(a) - key(b)).max()) if a.size else 0


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / den


def check_grad_samples(named_grads, names, values, offsets, tol):
    """Sampled entries of every gradient tensor (fixtures written by oracle/make_golden.grad_samples): the error of
    each tensor's samples relative to that tensor's largest sampled magnitude.  Returns the worst ratio."""
    from stylerenderer_amd import synth

    assert sorted(named_grads) == list(names)
    worst = 0.0
    for i, n in enumerate(names):
        want = np.asarray(values[offsets[i]:offsets[i + 1]], np.float64)
        g = named_grads[n].detach().reshape(-1).cpu().numpy().astype(np.float64)
        got = g[synth.sample_index(g.size, 256)]
        assert got.shape == want.shape, n
        scale = max(float(np.abs(want).max()), 1e-12)
        err = float(np.abs(got - want).max()) / scale
        assert err <= tol, "%s: sampled-gradient error %.3e of the tensor's scale (bar %.1e)" % (n, err, tol)
        worst = max(worst, err)
    return worst
