"""csrc/bank_mm.hip through op/bankmm.py: the table-driven skinny products of all modulated layers of a pass, their
gradients of first and second order against float64 tensor algebra, and the three forms of op.style_bank (native
tables, stacked library GEMMs, per layer) against each other on a real generator."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _mod_ref(x, rows, ws, bs, alpha, bscale):
    return [alpha * x[:, r] @ w.t() + bscale * b for r, w, b in zip(rows, ws, bs)]


def _problems(batch, seed=0):
    g = torch.Generator().manual_seed(seed)
    rows = [0, 1, 1, 2, 3, 3, 3, 5]                       # shared rows, an unused row (4), a gap
    widths = [64, 128, 32, 64, 512, 256, 8, 128]
    r, k = 6, 128
    x = torch.randn(batch, r, k, generator=g)
    ws = [torch.randn(n, k, generator=g) for n in widths]
    bs = [torch.randn(n, generator=g) for n in widths]
    return rows, widths, x, ws, bs


@pytest.mark.parametrize("batch", [1, 2, 4, 11])
def test_modulation_family_first_and_second_order(batch):
    from stylerenderer_amd.op import bankmm

    rows, widths, x, ws, bs = _problems(batch)
    alpha, bscale = 0.37, 1.5
    xd = x.to(DEV).requires_grad_(True)
    wd = [w.to(DEV).requires_grad_(True) for w in ws]
    bd = [b.to(DEV).requires_grad_(True) for b in bs]
    x64 = x.double().requires_grad_(True)
    w64 = [w.double().requires_grad_(True) for w in ws]
    b64 = [b.double().requires_grad_(True) for b in bs]
    assert bankmm.modulation_supported(xd, wd, bd)
    flat, outs = bankmm.modulation(xd, rows, wd, bd, alpha, bscale)
    refs = _mod_ref(x64, rows, w64, b64, alpha, bscale)
    for o, r in zip(outs, refs):
        assert o.is_contiguous() and o.data_ptr() % 16 == 0
        torch.testing.assert_close(o.double().cpu(), r.detach(), rtol=2e-6, atol=2e-5)
    # a nonlinear functional so that the second order is not trivially zero: sum_p <c_p, s_p^2> ... then the squared
    # norm of its latent gradient (what the path-length regulariser does, reference train.py:118-134)
    cs = [torch.randn(batch, n, generator=torch.Generator().manual_seed(7 + i)) for i, n in enumerate(widths)]
    f = sum(((o * o) * c.to(DEV)).sum() for o, c in zip(outs, cs))
    f64 = sum(((r * r) * c.double()).sum() for r, c in zip(refs, cs))
    gx, = torch.autograd.grad(f, xd, create_graph=True)
    gx64, = torch.autograd.grad(f64, x64, create_graph=True)
    torch.testing.assert_close(gx.double().cpu(), gx64.detach(), rtol=1e-5, atol=1e-5 * float(gx64.detach().abs().max()))
    assert float(gx[:, 4].abs().max()) == 0.0                     # the row nobody reads
    pen = (gx * gx).sum()
    pen64 = (gx64 * gx64).sum()
    got = torch.autograd.grad(pen, [xd] + wd + bd, retain_graph=True)
    want = torch.autograd.grad(pen64, [x64] + w64 + b64, retain_graph=True)
    for a, b_ in zip(got, want):
        torch.testing.assert_close(a.double().cpu(), b_, rtol=2e-5, atol=2e-5 * float(b_.abs().max()) + 1e-12)
    # first-order weight / bias gradients, and run-to-run identity (fixed summation order)
    g1 = torch.autograd.grad(f, wd + bd, retain_graph=True)
    g2 = torch.autograd.grad(f, wd + bd)
    w1 = torch.autograd.grad(f64, w64 + b64)
    for a, a2, b_ in zip(g1, g2, w1):
        assert torch.equal(a, a2)
        torch.testing.assert_close(a.double().cpu(), b_, rtol=1e-5, atol=1e-5 * float(b_.abs().max()) + 1e-12)


@pytest.mark.parametrize("batch", [1, 4])
def test_demodulation_family_first_and_second_order(batch):
    from stylerenderer_amd.op import bankmm

    g = torch.Generator().manual_seed(3)
    widths = [64, 32, 128, 64, 16]                       # blocks of the A flat; problems read blocks 0, 2, 3
    read = [0, 2, 3]
    couts = [128, 64, 256]
    a = torch.rand(sum(batch * w for w in widths), generator=g) + 0.1
    ms = [torch.rand(widths[i], co, generator=g) for i, co in zip(read, couts)]
    offs, o = [], 0
    for w in widths:
        offs.append(o)
        o += batch * w
    ad = a.to(DEV).requires_grad_(True)
    md = [m.to(DEV).requires_grad_(True) for m in ms]
    a64 = a.double().requires_grad_(True)
    m64 = [m.double().requires_grad_(True) for m in ms]
    assert bankmm.demod_supported(md)
    flat, qs = bankmm.demod_products(ad, batch, [offs[i] for i in read], md)
    blocks64 = [a64[offs[i]:offs[i] + batch * widths[i]].view(batch, widths[i]) for i in read]
    refs = [blk @ m for blk, m in zip(blocks64, m64)]
    for q, r in zip(qs, refs):
        torch.testing.assert_close(q.double().cpu(), r.detach(), rtol=2e-6, atol=1e-5)
    d = torch.rsqrt(flat + 1e-8)
    d64 = torch.cat([torch.rsqrt(r + 1e-8).reshape(-1) for r in refs])
    f = (d * torch.linspace(0.5, 1.5, d.numel(), device=DEV)).sum()
    f64 = (d64 * torch.linspace(0.5, 1.5, d64.numel(), dtype=torch.float64)).sum()
    ga, = torch.autograd.grad(f, ad, create_graph=True)
    ga64, = torch.autograd.grad(f64, a64, create_graph=True)
    torch.testing.assert_close(ga.double().cpu(), ga64.detach(), rtol=2e-5, atol=2e-5 * float(ga64.detach().abs().max()))
    lo, hi = offs[1], offs[1] + batch * widths[1]
    assert float(ga[lo:hi].abs().max()) == 0.0                     # a block no problem reads
    pen = (ga * ga).sum()
    pen64 = (ga64 * ga64).sum()
    got = torch.autograd.grad(pen, [ad] + md)
    want = torch.autograd.grad(pen64, [a64] + m64)
    for x, y in zip(got, want):
        torch.testing.assert_close(x.double().cpu(), y, rtol=1e-4, atol=1e-4 * float(y.abs().max()) + 1e-12)


def test_bank_entry_points_reject_bad_tables():
    import ctypes

    from stylerenderer_amd import _lib

    L = _lib.lib()
    x = torch.zeros(2, 8, device=DEV)
    w = torch.zeros(4, 8, device=DEV)
    o = torch.zeros(2, 4, device=DEV)
    one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())            # noqa: E731
    i64 = lambda v: (ctypes.c_int64 * 1)(v)                        # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    assert L.sr_bank_nt(1, one(o), one(x), one(w), None, i64(8), i64(4), i64(8), i64(4), 2, 1.0, 0.0, st) == 0
    assert L.sr_bank_nt(1, one(o), one(x), one(w), None, i64(8), i64(4), i64(6), i64(4), 2, 1.0, 0.0, st) != 0   # K % 4
    assert L.sr_bank_nt(1, one(o), one(x), one(w), None, i64(4), i64(4), i64(8), i64(4), 2, 1.0, 0.0, st) != 0   # lda < K
    assert L.sr_bank_nt(1, None, one(x), one(w), None, i64(8), i64(4), i64(8), i64(4), 2, 1.0, 0.0, st) != 0
    torch.cuda.synchronize()


def _generator(size=64, seed=0):
    from stylerenderer_amd import model, synth

    torch.manual_seed(seed)
    g = model.Generator(size, 512, 8).to(DEV)
    sd = g.state_dict()
    synth.fill_state_dict(sd, seed + 1)
    g.load_state_dict(sd)
    return g


def _linear_activations(g):
    """slope 1: every LeakyReLU linear, so that two fp32 evaluation orders cannot land a pre-activation on different sides
    of the kink (one such element moves every upstream gradient by ~1e-3: tests/test_model_gpu.py)."""
    from stylerenderer_amd.op import FusedLeakyReLU

    for m in g.modules():
        if isinstance(m, FusedLeakyReLU):
            m.negative_slope = 1.0
    return g


@pytest.mark.parametrize("batch", [1, 4])
def test_style_bank_forms_agree_on_a_generator(batch, monkeypatch):
    """Native tables == stacked library GEMMs == per-layer kernels: image, latent gradient, every parameter gradient, and
    the path-length double backward (the stacked form needs rocBLAS: SR_STRICT_NATIVE off for that leg only); linear
    activations, so the three are compared as the same algebra in three summation orders."""
    from stylerenderer_amd import train

    g = _linear_activations(_generator())
    z = torch.randn(batch, 512, device=DEV, generator=torch.Generator(DEV).manual_seed(5))
    noise = [n.detach() for n in g.make_noise()]
    params = [p for p in g.parameters() if p.requires_grad]

    def run(mode):
        monkeypatch.setenv("SR_STYLE_BANK", mode)
        monkeypatch.setenv("SR_STRICT_NATIVE", "0" if mode == "stacked" else "1")
        w = g.style(z).unsqueeze(1).repeat(1, g.n_latent, 1).detach().requires_grad_(True)
        img, _ = g([w], input_is_latent=True, noise=noise)
        probe = torch.randn(img.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(11))
        loss, mean, lengths = train.g_path_regularize(img, w, torch.zeros((), device=DEV), noise=probe)
        grads = torch.autograd.grad(loss + img.square().mean(), [w] + params, allow_unused=True)
        return img.detach(), lengths.detach(), grads

    run("1")                       # (the first pass of a network records its weight-bank plan layer by layer)
    ref = run("1")
    for mode in ("stacked", "0"):
        got = run(mode)
        torch.testing.assert_close(got[0], ref[0], rtol=0, atol=2e-5 * float(ref[0].abs().max()))
        torch.testing.assert_close(got[1], ref[1], rtol=2e-5, atol=0)
        for a, b_ in zip(got[2], ref[2]):
            if a is None or b_ is None:
                assert a is None and b_ is None
                continue
            # (the two older forms differ by up to 1.1e-4 of a tensor's scale between themselves on the one-element noise
            # strengths — sums of ~1e5 signed terms —, 1.5e-5 elsewhere: scripts/bank_cmp_probe.py)
            bar = 3e-4 if b_.numel() == 1 else 5e-5
            torch.testing.assert_close(a, b_, rtol=0, atol=bar * float(b_.abs().max()) + 1e-12)
    names = ["latent"] + [n for n, p in g.named_parameters() if p.requires_grad]
    for mode in ("1", "0"):
        first, again = run(mode), run(mode)
        moved = [(n, float((a - b_).abs().max())) for n, a, b_ in zip(names, again[2], first[2])
                 if a is not None and not torch.equal(a, b_)]
        assert torch.equal(again[0], first[0]) and not moved, (mode, moved[:8])


def test_native_bank_issues_no_library_gemm(monkeypatch):
    """With the table kernels on, a generator forward + backward under SR_STRICT_NATIVE dispatches no aten::bmm /
    baddbmm / mm / addmm ("no device tensor leaves this library" for the style path)."""
    from torch.utils._python_dispatch import TorchDispatchMode

    monkeypatch.setenv("SR_STYLE_BANK", "1")
    g = _generator()
    z = torch.randn(2, 512, device=DEV)
    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func._schema.name
            if name in ("aten::bmm", "aten::baddbmm", "aten::mm", "aten::addmm", "aten::matmul", "aten::linear"):
                seen.append(name)
            return func(*args, **(kwargs or {}))

    with Spy():
        img, _ = g([z])
        img.square().mean().backward()
    assert not seen, seen
