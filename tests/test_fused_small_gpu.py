"""Round-6 launch-count fusions, each against the operators it replaces: ToRGB's modulated weight rows (sr_modrows_*), the
multi-resolution rasterizer node with in-kernel gradient accumulation (SR_RASTER_GRAD_ACC), the row-dot with its division
(sr_rowdot_div), the pixel loss (sr_mse_*), and bias + ReLU of the LPIPS trunk in the Winograd store."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.mark.parametrize("b,n,c", [(1, 3, 512), (4, 3, 128), (5, 4, 36)])
def test_modulated_rows_match_the_tensor_products(b, n, c):
    from stylerenderer_amd.op import smallconv

    g = torch.Generator().manual_seed(b * 100 + c)
    w = torch.randn(n, c, generator=g).to(DEV).requires_grad_(True)
    s = torch.randn(b, c, generator=g).to(DEV).requires_grad_(True)
    up = torch.randn(b, n, c, generator=g).to(DEV)
    scale = 0.173
    got = smallconv.modulated_rows(w, s, scale)
    want = (w * scale)[None] * s[:, None]
    assert torch.equal(got, want)                                            # same two multiplications per element
    gw, gs = torch.autograd.grad(got, [w, s], up)
    ww, ws = torch.autograd.grad(want, [w, s], up)
    torch.testing.assert_close(gw, ww, rtol=2e-6, atol=2e-6 * float(ww.abs().max()))
    torch.testing.assert_close(gs, ws, rtol=2e-6, atol=2e-6 * float(ws.abs().max()))
    # recorded backward (create_graph): second order through the node equals second order through the products
    w2 = w.detach().clone().requires_grad_(True)
    s2 = s.detach().clone().requires_grad_(True)
    a, = torch.autograd.grad(smallconv.modulated_rows(w, s, scale), s, up, create_graph=True)
    b_, = torch.autograd.grad((w2 * scale)[None] * s2[:, None], s2, up, create_graph=True)
    ga, = torch.autograd.grad(a.square().sum(), w)
    gb, = torch.autograd.grad(b_.square().sum(), w2)
    torch.testing.assert_close(ga, gb, rtol=1e-5, atol=1e-5 * float(gb.abs().max()))


def _posed_mesh(batch):
    from stylerenderer_amd import synth

    v0, tri = synth.uv_ellipsoid(24, 20)
    v = torch.from_numpy(synth.random_poses(v0, batch, seed=3)).to(DEV)
    nrm = torch.from_numpy(synth.vertex_normals(v.cpu().numpy(), tri)).to(DEV)
    return v, nrm, torch.from_numpy(tri).to(DEV)


@pytest.mark.parametrize("batch", [1, 3])
def test_rasterize_pyramid_equals_separate_calls(batch, monkeypatch):
    from stylerenderer_amd.op.rasterize import rasterize, rasterize_pyramid

    v, nrm, tri = _posed_mesh(batch)
    sizes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)]
    ups = [torch.randn(batch, 3, h, w, device=DEV, generator=torch.Generator(DEV).manual_seed(h)) for h, w in sizes]

    def run(pyramid):
        vv, nn = v.clone().requires_grad_(True), nrm.clone().requires_grad_(True)
        if pyramid:
            maps = rasterize_pyramid(vv, nn, tri, sizes, channel_major=True)
        else:
            maps = [rasterize(vv, nn, tri, h, w, channel_major=True) for h, w in sizes]
        f = sum((m * u).sum() for m, u in zip(maps, ups))
        gv, gn = torch.autograd.grad(f, [vv, nn])
        return [m.detach() for m in maps], gv, gn

    maps_p, gv_p, gn_p = run(True)
    maps_s, gv_s, gn_s = run(False)
    for a, b_ in zip(maps_p, maps_s):
        assert torch.equal(a, b_)
    # the same per-resolution gradients, added in the gather kernel (in list order) instead of by autograd
    torch.testing.assert_close(gv_p, gv_s, rtol=0, atol=2e-6 * float(gv_s.abs().max()))
    torch.testing.assert_close(gn_p, gn_s, rtol=0, atol=2e-6 * float(gn_s.abs().max()))
    again = run(True)
    assert torch.equal(again[1], gv_p) and torch.equal(again[2], gn_p)          # deterministic
    # a map nobody differentiates: its gradient pass is skipped, the others still add up
    vv, nn = v.clone().requires_grad_(True), nrm.clone().requires_grad_(True)
    maps = rasterize_pyramid(vv, nn, tri, sizes, channel_major=True)
    gv, = torch.autograd.grad((maps[1] * ups[1]).sum() + (maps[3] * ups[3]).sum(), vv)
    v2 = v.clone().requires_grad_(True)
    ref = sum((rasterize(v2, nrm, tri, *sizes[k], channel_major=True) * ups[k]).sum() for k in (1, 3))
    gr, = torch.autograd.grad(ref, v2)
    torch.testing.assert_close(gv, gr, rtol=0, atol=2e-6 * float(gr.abs().max()))
    monkeypatch.setenv("SR_RASTER_PYRAMID", "0")
    off = run(True)
    assert torch.equal(off[1], gv_s)
    monkeypatch.delenv("SR_RASTER_PYRAMID")
    # the three-launch forward over all levels (sr_rasterize_forward_levels_f32) against the per-level launches: the maps,
    # and the gradient state they leave (same gradients, bit for bit)
    # ... and the four-launch gradient over all levels (sr_rasterize_grad_levels_f32) against the accumulating per-level
    # calls (SR_RASTER_GRAD_ACC): the same sums in the same order
    monkeypatch.setenv("SR_RASTER_LEVELS", "0")
    per_level = run(True)
    for a, b_ in zip(per_level[0], maps_p):
        assert torch.equal(a, b_)
    assert torch.equal(per_level[1], gv_p) and torch.equal(per_level[2], gn_p)


def test_rasterize_levels_entry_point_matches_single_level_calls():
    """C ABI level: attribute maps and winner maps of sr_rasterize_forward_levels_f32 == sr_rasterize_forward_f32 per level,
    and the refusal of a level the dispatcher gives to the LDS-tiled path."""
    import ctypes

    import importlib

    from stylerenderer_amd import _lib

    R = importlib.import_module("stylerenderer_amd.op.rasterize")          # (op.rasterize the attribute is the function)
    v, nrm, tri = _posed_mesh(2)
    sizes = [(8, 8), (16, 16), (24, 40), (64, 64)]
    got, states = R._forward_levels(v, nrm, tri, sizes, False, 1e-6, True, True)
    assert got is not None
    for k, (h, w) in enumerate(sizes):
        _, _, _, ref, st = R._forward_impl(v, tri, h, w, False, 1e-6, tex=nrm, want_index=False, want_win=True, chw=True)
        assert torch.equal(got[k], ref)
        assert torch.equal(states[2 * k], st[0])                         # winner maps
    L = _lib.lib()
    one = lambda x: (ctypes.c_int64 * 1)(x)                              # noqa: E731
    assert L.sr_rasterize_levels_supported(1, 2, tri.shape[0], one(64), one(64)) == 1
    assert L.sr_rasterize_levels_supported(1, 64, tri.shape[0], one(256), one(256)) == 0     # 4 096 tiles: the tiled path


@pytest.mark.parametrize("shape", [(1, 64, 256, 256), (3, 5, 6, 10)])
def test_maxpool2_matches_torch_including_ties(shape):
    from stylerenderer_amd.op.lpips_layer import max_pool2

    g = torch.Generator(DEV).manual_seed(9)
    x = torch.relu(torch.randn(shape, device=DEV, generator=g))            # ReLU output: windows full of equal zeros
    x[0, 0, 0, 0] = float("nan")
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = max_pool2(xa)
    yb = torch.nn.functional.max_pool2d(xb, 2, 2)
    assert torch.equal(torch.nan_to_num(ya.detach(), nan=-7.0), torch.nan_to_num(yb.detach(), nan=-7.0))
    up = torch.randn(ya.shape, device=DEV, generator=g)
    ga, = torch.autograd.grad(ya, xa, up)
    gb, = torch.autograd.grad(yb, xb, up)
    assert torch.equal(ga, gb)


@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 64, 128, 128), (3, 5, 9, 9)])
def test_rowdot_div_equals_rowdot_then_divide(shape):
    from stylerenderer_amd.op.fused_elem import rowdot, rowdot_div

    g = torch.Generator(DEV).manual_seed(sum(shape))
    a = torch.randn(shape, device=DEV, generator=g)
    b_ = torch.randn(shape, device=DEV, generator=g)
    d = torch.rand(shape[:2], device=DEV, generator=g) + 0.5
    with torch.no_grad():
        got = rowdot_div(a, b_, d)
        want = rowdot(a, b_) / d
    assert torch.equal(got, want)
    # with a graph being recorded the composite runs, differentiable
    a2 = a.clone().requires_grad_(True)
    out = rowdot_div(a2, b_, d)
    assert out.requires_grad
    torch.testing.assert_close(out.detach(), want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n", [(1, 3, 256, 256), (2, 3, 17, 5)])
def test_mse_kernels(n):
    from stylerenderer_amd.op.lpips_layer import mse

    g = torch.Generator(DEV).manual_seed(1)
    a = torch.randn(n, device=DEV, generator=g).requires_grad_(True)
    b_ = torch.randn(n, device=DEV, generator=g)
    got = mse(a, b_)
    want = torch.mean((a.double() - b_.double()) ** 2)
    torch.testing.assert_close(got.double(), want.detach(), rtol=2e-6, atol=0)
    ga, = torch.autograd.grad(got * 3.0, a)
    gw, = torch.autograd.grad(torch.mean((a - b_) ** 2) * 3.0, a)
    torch.testing.assert_close(ga, gw, rtol=2e-6, atol=1e-9)
    assert torch.equal(mse(a, b_), got)


def test_lpips_trunk_fused_bias_relu_is_bit_identical(monkeypatch):
    from stylerenderer_amd import lpips

    net = lpips.PNetLin().to(DEV)
    x = (torch.rand(1, 3, 256, 256, device=DEV, generator=torch.Generator(DEV).manual_seed(2)) * 2 - 1).requires_grad_(True)
    t = torch.rand(1, 3, 256, 256, device=DEV, generator=torch.Generator(DEV).manual_seed(3)) * 2 - 1
    with torch.no_grad():
        feats = net.features(t)

    def run(flag):
        monkeypatch.setenv("SR_LPIPS_FUSED_TRUNK", flag)
        d = net.distance_to(feats, x).sum()
        g, = torch.autograd.grad(d, x)
        return d.detach(), g

    d1, g1 = run("1")
    d0, g0 = run("0")
    assert torch.equal(d1, d0)
    assert torch.equal(g1, g0)
