"""GPU parity tests: HIP kernels (through the C ABI and the op wrappers) against the CPU oracle
and the golden vectors generated from the reference."""
import importlib

import numpy as np
import pytest
import torch

import ops_np
import raster
from make_golden_cases import UFD_TAGS
from util import bits_equal, max_ulp, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a, **kw):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, **kw)


# ------------------------------------------------------------------------------- fused bias act
@pytest.mark.parametrize("shape", [(2, 4, 5, 5), (3, 8), (2, 3, 7), (4, 16, 32, 32), (1, 5, 33, 31),
                                   (16, 512), (2, 8, 64, 64)])
@pytest.mark.parametrize("code", [(3, 0), (3, 1), (3, 2), (1, 0)])
def test_fused_bias_act_bitexact_vs_oracle(shape, code):
    from stylerenderer_amd import synth
    from stylerenderer_amd.op.fused_act import fused_bias_act

    act, grad = code
    x = synth.det_normal(shape, 5)
    x.reshape(-1)[::11] = 0
    b = synth.det_normal((shape[1],), 6)
    ref = synth.det_normal(shape, 7)
    for use_b in (True, False):
        y = fused_bias_act(T(x), T(b) if use_b else torch.empty(0, device=DEV),
                           T(ref) if grad == 1 else torch.empty(0, device=DEV), act, grad, 0.2, 2 ** 0.5)
        want = ops_np.fused_bias_act(x, b if use_b else None, ref if grad == 1 else None, act, grad)
        assert bits_equal(y.cpu().numpy(), want)


def test_fused_bias_act_unaligned_views():
    from stylerenderer_amd import synth
    from stylerenderer_amd.op.fused_act import fused_bias_act

    base = T(synth.det_normal((4 * 6 * 10 * 10 + 3,), 8))
    x = base[3:].view(4, 6, 10, 10)                      # data_ptr only 4-byte aligned
    b = T(synth.det_normal((6,), 9))
    y = fused_bias_act(x, b, torch.empty(0, device=DEV), 3, 0, 0.2, 2 ** 0.5)
    want = ops_np.fused_bias_act(x.cpu().numpy(), b.cpu().numpy(), None, 3, 0)
    assert bits_equal(y.cpu().numpy(), want)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_fused_leaky_relu_autograd_vs_golden(golden, tag):
    import stylerenderer_amd.op as op

    g = golden("fused_act")
    x = T(g[tag + "_x"]).requires_grad_()
    b = T(g[tag + "_bias"]).requires_grad_()
    y = op.fused_leaky_relu(x, b)
    assert bits_equal(y.detach().cpu().numpy(), g[tag + "_y"])
    gy = T(g[tag + "_gy"]).requires_grad_()
    gx, gb = torch.autograd.grad(y, [x, b], gy, create_graph=True)
    assert max_ulp(gx.detach().cpu().numpy(), g[tag + "_gx"]) <= 2
    assert np.allclose(gb.detach().cpu().numpy(), g[tag + "_gb"], rtol=1e-5, atol=1e-6)
    (ggo,) = torch.autograd.grad((gx * T(g[tag + "_ggx"])).sum() + (gb * T(g[tag + "_ggb"])).sum(), gy)
    assert max_ulp(ggo.cpu().numpy(), g[tag + "_ggo"]) <= 2


@pytest.mark.parametrize("shape", [(4, 16, 32, 32), (2, 8, 64, 64), (3, 5, 40, 36), (16, 512), (2, 3, 7, 9)])
def test_fused_act_backward_bias_reduction(shape):
    from stylerenderer_amd import synth
    from stylerenderer_amd.op.fused_act import _act_backward

    gy = synth.det_normal(shape, 15)
    out = synth.det_normal(shape, 16)
    gx, gb = _act_backward(T(gy), T(out), 0.2, 2 ** 0.5)
    wx, wb = ops_np.fused_leaky_relu_backward(gy, out)
    assert bits_equal(gx.cpu().numpy(), wx)
    assert np.allclose(gb.cpu().numpy(), wb, rtol=2e-5, atol=1e-5)
    gx2, gb2 = _act_backward(T(gy), T(out), 0.2, 2 ** 0.5)     # deterministic reduction
    assert bits_equal(gb.cpu().numpy(), gb2.cpu().numpy())


def test_fused_module_state_and_cpu_dispatch():
    import stylerenderer_amd.op as op

    m = op.FusedLeakyReLU(6)
    assert list(m.state_dict()) == ["bias"]
    x = torch.randn(2, 6, 4, 4)
    y_cpu = m(x)
    y_gpu = m.to(DEV)(x.to(DEV))
    assert torch.allclose(y_cpu, y_gpu.cpu(), atol=1e-6)


# ------------------------------------------------------------------------------- upfirdn2d
@pytest.mark.parametrize("tag", UFD_TAGS)
def test_upfirdn2d_vs_oracle_and_golden(golden, tag):
    import stylerenderer_amd.op as op

    g = golden("upfirdn2d")
    up, down, p0, p1 = [int(t) for t in g[tag + "_prm"]]
    x = T(g[tag + "_x"]).requires_grad_()
    k = T(g[tag + "_k"])
    y = op.upfirdn2d(x, k, up=up, down=down, pad=(p0, p1))
    want = ops_np.upfirdn2d(g[tag + "_x"], g[tag + "_k"], up, down, (p0, p1))
    assert np.array_equal(y.detach().cpu().numpy(), want)           # bit-exact vs the oracle (-0 == +0)
    assert np.abs(y.detach().cpu().numpy() - g[tag + "_y"]).max() <= 4e-7 * max(1, np.abs(g[tag + "_y"]).max())
    gy = T(g[tag + "_gy"]).requires_grad_()
    (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
    want_gx = ops_np.upfirdn2d_backward(g[tag + "_gy"], g[tag + "_k"], g[tag + "_x"].shape, up, down, (p0, p1))
    assert np.array_equal(gx.detach().cpu().numpy(), want_gx)
    assert np.abs(gx.detach().cpu().numpy() - g[tag + "_gx"]).max() <= 4e-7 * max(1, np.abs(g[tag + "_gx"]).max())
    # double backward = the forward operator applied to the grad-of-grad
    (ggy,) = torch.autograd.grad(gx, gy, x.detach())
    assert np.array_equal(ggy.cpu().numpy(), want)


@pytest.mark.parametrize("res,pad", [(64, (1, 1)), (64, (2, 2)), (128, (1, 1))])
def test_upfirdn2d_blur_generator_shapes(res, pad):
    """Blur after the stride-2 transposed conv: [B, C, r+1, r+1] -> [B, C, r, r] (reference layers.py:272-275)
    and its gradient shape, against the oracle."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth

    k = ops_np.make_blur_kernel((1, 3, 3, 1), 4.0)
    n_in = res + 1 if pad == (1, 1) else res
    x = synth.det_normal((2, 5, n_in, n_in), 31)
    y = op.upfirdn2d(T(x), T(k), pad=pad)
    assert np.array_equal(y.cpu().numpy(), ops_np.upfirdn2d(x, k, 1, 1, pad))


def test_upfirdn2d_full_size_properties():
    """BASELINE config size (B=16, C=128, 257 -> 256): linearity and an adjoint identity
    <blur(x), y> == <x, blur^T(y)> instead of a CPU comparison."""
    import stylerenderer_amd.op as op

    k = T(ops_np.make_blur_kernel((1, 3, 3, 1), 4.0))
    gen = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(16, 128, 257, 257, device=DEV, generator=gen, requires_grad=True)
    y = op.upfirdn2d(x, k, pad=(1, 1))
    assert y.shape == (16, 128, 256, 256)
    w = torch.randn(y.shape, device=DEV, generator=gen)
    (gx,) = torch.autograd.grad(y, x, w)
    lhs = (y.double() * w.double()).sum().item()
    rhs = (x.detach().double() * gx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0)
    # spot-check two planes against the oracle
    xs = x.detach()[3:4, 7:9].cpu().numpy()
    assert np.array_equal(y.detach()[3:4, 7:9].cpu().numpy(),
                          ops_np.upfirdn2d(xs, k.cpu().numpy(), 1, 1, (1, 1)))


# ------------------------------------------------------------------------------- rasterizer
RASTER_CASES = ["raster_ellipsoid_32", "raster_ellipsoid_64", "raster_perspective_32", "raster_adversarial_16"]


@pytest.fixture(params=["auto", "tiled"])
def raster_path(request, monkeypatch):
    """The forward has two device paths chosen per call (csrc/rasterize.hip tiled_ok): the LDS-tiled one only pays for
    thousands of (sample, tile) workgroups, so small test inputs would never reach it — every bit-exactness test runs
    once with the automatic choice and once with the tiled path forced."""
    if request.param == "tiled":
        monkeypatch.setenv("SR_RASTER_TILED", "1")
    else:
        monkeypatch.delenv("SR_RASTER_TILED", raising=False)
    return request.param



@pytest.mark.parametrize("name", RASTER_CASES)
def test_rasterize_bitexact_vs_golden(golden, name, raster_path):
    R = importlib.import_module("stylerenderer_amd.op.rasterize")

    g = golden(name)
    persp = "perspective" in name
    res = g["index"].shape[1]
    v, tri = T(g["v"]), T(g["tri"].astype(np.int64))
    idx, coeff, zbuf = R.forward_with_depth(v, tri, res, res, persp, 1e-6)
    assert np.array_equal(idx.cpu().numpy(), g["index"].astype(np.int64))
    assert bits_equal(coeff.cpu().numpy(), g["coeff"])
    assert bits_equal(zbuf.cpu().numpy(), g["zbuf"])
    d = R.backward(v, idx, persp, 1e-6)
    assert bits_equal(d.cpu().numpy(), g["dcoeff"])
    # pybind-style entry returns the same pair
    i2, c2 = R.rasterize_op.forward(v, tri, res, 0, persp, 1e-6)
    assert torch.equal(i2, idx) and torch.equal(c2, coeff)


def test_rasterize_misc_topologies(golden, raster_path):
    R = importlib.import_module("stylerenderer_amd.op.rasterize")

    g = golden("raster_misc")
    idx, coeff, zbuf = R.forward_with_depth(T(g["v"]), T(g["tri_b"].astype(np.int64)), 16, 16, False, 1e-6)
    assert np.array_equal(idx.cpu().numpy(), g["index_b"].astype(np.int64))
    assert bits_equal(coeff.cpu().numpy(), g["coeff_b"]) and bits_equal(zbuf.cpu().numpy(), g["zbuf_b"])
    idx, coeff = R.forward(T(g["v_bf"]), T(g["tri_bf"].astype(np.int64)), 32, 32, False, 1e-6)
    assert np.array_equal(idx.cpu().numpy(), g["index_bf"].astype(np.int64))
    assert bits_equal(coeff.cpu().numpy(), g["coeff_bf"])


def test_rasterize_known_answer_fp64_and_gradcheck(golden, raster_path):
    """The reference's __main__ test (reference op/rasterize.py:83-107) on the HIP path."""
    import stylerenderer_amd.op as op
    R = importlib.import_module("stylerenderer_amd.op.rasterize")

    g = golden("raster_kat")
    v = T(g["v"]).requires_grad_()
    t = T(g["tex"]).requires_grad_()
    f = T(g["f"])
    o = op.rasterize(v, t, f, 5)
    assert bits_equal(o.detach().cpu().numpy(), g["out"])
    idx, coeff = R.forward(v.detach(), f, 5, 0, False, 1e-6)
    assert np.array_equal(idx.cpu().numpy(), g["index"].astype(np.int64))
    assert bits_equal(coeff.cpu().numpy(), g["coeff"])
    assert bits_equal(R.backward(v.detach(), idx, False, 1e-6).cpu().numpy(), g["dcoeff"])
    gv, gt = torch.autograd.grad(o, [v, t], T(g["grad_out"]))
    assert np.allclose(gv.cpu().numpy(), g["grad_v"], rtol=2e-6, atol=1e-7)
    assert np.allclose(gt.cpu().numpy(), g["grad_tex"], rtol=2e-6, atol=1e-7)
    cat = torch.cat((v.detach(), t.detach()), -1).requires_grad_()
    assert torch.autograd.gradcheck(lambda x: op.rasterize(x[:, :, :3], x[:, :, 3:], f, 5), cat,
                                    eps=1e-6, atol=1e-6, nondet_tol=1e-9)


@pytest.mark.parametrize("res", [32, 64])
def test_rasterize_interp_and_grads_vs_golden(golden, res, raster_path):
    import stylerenderer_amd.op as op

    g = golden("raster_ellipsoid_%d" % res)
    v = T(g["v"]).requires_grad_()
    tex = T(g["tex"]).requires_grad_()
    tri = T(g["tri"].astype(np.int64))
    out = op.rasterize(v, tex, tri, res)
    want = raster.rasterize(g["v"], g["tex"], g["tri"].astype(np.int64), res)
    assert bits_equal(out.detach().cpu().numpy(), want)                 # oracle: bitwise
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() <= 2e-7  # reference torch.sum order
    gv, gt = torch.autograd.grad(out, [v, tex], T(g["grad_out"]))
    assert np.abs(gv.cpu().numpy() - g["grad_v"]).max() <= 2e-5 * np.abs(g["grad_v"]).max()
    assert np.abs(gt.cpu().numpy() - g["grad_tex"]).max() <= 2e-6 * np.abs(g["grad_tex"]).max()


def test_rasterize_face_mesh_256_vs_oracle_and_determinism(raster_path):
    """BFM-size-class mesh (24 770 vertices, 49 536 triangles) at 256x256: bitwise equal to the C
    oracle on the same inputs, and byte-identical across repeated launches (the reference's CUDA
    kernel is racy)."""
    from stylerenderer_amd import synth
    R = importlib.import_module("stylerenderer_amd.op.rasterize")

    v0, tri = synth.face_sized_mesh()
    v = synth.random_poses(v0, 3, seed=11)
    idx, coeff, zbuf = R.forward_with_depth(T(v), T(tri), 256, 256, False, 1e-6)
    wi, wc, wz = raster.forward_buffers(v, tri, 256, 256, False, 1e-6)
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert bits_equal(coeff.cpu().numpy(), wc) and bits_equal(zbuf.cpu().numpy(), wz)
    cover = float((wi != 0).any(-1).mean())
    assert 0.2 < cover < 0.9
    for _ in range(3):
        i2, c2, z2 = R.forward_with_depth(T(v), T(tri), 256, 256, False, 1e-6)
        assert torch.equal(i2, idx) and bits_equal(c2.cpu().numpy(), coeff.cpu().numpy())
    d = R.backward(T(v), idx, False, 1e-6)
    assert bits_equal(d.cpu().numpy(), raster.backward_dcoeff(v, wi, False, 1e-6))


def test_rasterize_batch64_bitwise_vs_oracle(raster_path):
    """BASELINE config[3] in full (B=64, 49 536 triangles, 256x256): ALL 64 samples — triangle ids, barycentric
    weights, z-buffer and the backward's dcoeff — bit for bit against the C oracle (41 Mtri/s: 77 ms for the batch),
    plus per-sample independence (a sample of the batch equals the same sample rasterised alone)."""
    from stylerenderer_amd import synth
    R = importlib.import_module("stylerenderer_amd.op.rasterize")

    v0, tri = synth.face_sized_mesh()
    vh = synth.random_poses(v0, 64, seed=5)
    v = T(vh)
    t = T(tri)
    idx, coeff, zbuf = R.forward_with_depth(v, t, 256, 256, False, 1e-6)
    wi, wc, wz = raster.forward_buffers(vh, tri, 256, 256, False, 1e-6)
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert bits_equal(coeff.cpu().numpy(), wc) and bits_equal(zbuf.cpu().numpy(), wz)
    i0, c0 = R.forward(v, t, 256, 256, False, 1e-6)
    assert torch.equal(i0, idx) and torch.equal(c0, coeff)
    d = R.backward(v, idx, False, 1e-6)
    assert bits_equal(d.cpu().numpy(), raster.backward_dcoeff(vh, wi, False, 1e-6))
    nv = v0.shape[0]
    for s in (0, 17, 63):
        i1, c1 = R.forward(v[s:s + 1].contiguous(), t, 256, 256, False, 1e-6)
        cov = (i1 != 0).any(-1)
        assert torch.equal(torch.where(cov[..., None], i1 + nv * s, i1), idx[s:s + 1])
        assert torch.equal(c1, coeff[s:s + 1])


def _oracle_grads(v, tex, tri, go, res):
    return raster.rasterize_grads(v, tex, tri, go, res)


def test_rasterize_gradients_are_deterministic_and_match_oracle_256(raster_path):
    """BFM-size-class mesh at 256x256: the two-phase gather (no float atomics) gives byte-identical grad_v /
    grad_tex on every launch, and agrees with the float64-accumulated oracle."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth

    v0, tri = synth.face_sized_mesh()
    vh = synth.random_poses(v0, 2, seed=21)
    nh = synth.vertex_normals(vh, tri)
    go = synth.det_normal((2, 256, 256, 3), 77)
    t = T(tri)
    runs = []
    for _ in range(3):
        v, n = T(vh).requires_grad_(), T(nh).requires_grad_()
        out = op.rasterize(v, n, t, 256)
        runs.append(torch.autograd.grad(out, [v, n], T(go)))
    for gv, gt in runs[1:]:
        assert torch.equal(gv, runs[0][0]) and torch.equal(gt, runs[0][1])
    wv, wt = _oracle_grads(vh, nh, tri, go, 256)
    assert np.abs(runs[0][0].cpu().numpy() - wv).max() <= 2e-5 * np.abs(wv).max()
    assert np.abs(runs[0][1].cpu().numpy() - wt).max() <= 2e-6 * np.abs(wt).max()


@pytest.mark.parametrize("c", [1, 2, 3, 6])
def test_rasterize_large_triangles_forward_and_gradients(c, raster_path):
    """A 218-triangle mesh at 256x256: every bounding box exceeds 64 pixels, so the workgroup-cooperative
    paths of k_depth_keys and k_grad_tri run; forward bitwise vs the C oracle, gradients vs the oracle, any
    attribute width (register accumulators are chunked by 4 channels), run-to-run identical."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth
    R = importlib.import_module("stylerenderer_amd.op.rasterize")

    v0, tri = synth.uv_ellipsoid(10, 12)
    vh = synth.random_poses(v0, 3, seed=31)
    tex = synth.det_normal((3, v0.shape[0], c), 32)
    idx, coeff, zbuf = R.forward_with_depth(T(vh), T(tri), 256, 256, False, 1e-6)
    wi, wc, wz = raster.forward_buffers(vh, tri, 256, 256, False, 1e-6)
    assert np.array_equal(idx.cpu().numpy(), wi) and bits_equal(coeff.cpu().numpy(), wc)
    assert bits_equal(zbuf.cpu().numpy(), wz)
    go = synth.det_normal((3, 256, 256, c), 33)
    got = []
    for _ in range(2):
        v, tx = T(vh).requires_grad_(), T(tex).requires_grad_()
        out = op.rasterize(v, tx, T(tri), 256)
        assert bits_equal(out.detach().cpu().numpy(), raster.rasterize(vh, tex, tri, 256))
        got.append(torch.autograd.grad(out, [v, tx], T(go)))
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    wv, wt = _oracle_grads(vh, tex, tri, go, 256)
    assert np.abs(got[0][0].cpu().numpy() - wv).max() <= 5e-5 * np.abs(wv).max()
    assert np.abs(got[0][1].cpu().numpy() - wt).max() <= 5e-6 * np.abs(wt).max()


@pytest.mark.parametrize("fwd,bwd", [("1", "0"), ("0", "1")])
def test_rasterize_gradient_follows_the_forward_record_not_the_environment(fwd, bwd, monkeypatch):
    """the gradient pass used to re-evaluate tiled_ok() (SR_RASTER_TILED + a size heuristic) to decide
    where the leader table lives; a different answer at backward time read a never-initialised table.  The forward now
    leaves a state word behind the table (1 = built by the tiled path, 0 = filled only) and the gradient pass reads IT
    on the device: flipping the environment between forward and backward changes nothing, bit for bit."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth

    v0, tri = synth.uv_ellipsoid(20, 18)
    vh = synth.random_poses(v0, 3, seed=9)
    nh = synth.vertex_normals(vh, tri)
    go = T(synth.det_normal((3, 64, 64, 3), 78))
    res = {}
    for tag, (e_f, e_b) in {"same": (fwd, fwd), "flipped": (fwd, bwd)}.items():
        v, n = T(vh).requires_grad_(), T(nh).requires_grad_()
        monkeypatch.setenv("SR_RASTER_TILED", e_f)
        out = op.rasterize(v, n, T(tri), 64)
        monkeypatch.setenv("SR_RASTER_TILED", e_b)
        res[tag] = torch.autograd.grad(out, [v, n], go, retain_graph=True)
        again = torch.autograd.grad(out, [v, n], go)                  # a second backward over the same state
        assert torch.equal(again[0], res[tag][0]) and torch.equal(again[1], res[tag][1])
    assert torch.equal(res["same"][0], res["flipped"][0]) and torch.equal(res["same"][1], res["flipped"][1])
    wv, wt = _oracle_grads(vh, nh, tri, go.cpu().numpy(), 64)
    assert np.abs(res["flipped"][0].cpu().numpy() - wv).max() <= 2e-5 * np.abs(wv).max()
    assert np.abs(res["flipped"][1].cpu().numpy() - wt).max() <= 2e-6 * np.abs(wt).max()


@pytest.mark.parametrize("tiled,tex_c,dtype", [("1", 3, "f32"), ("0", 3, "f32"), ("1", 5, "f32"), ("1", 3, "f64")])
def test_rasterize_channel_major_equals_the_permuted_output(tiled, tex_c, dtype, monkeypatch):
    """SR_RASTER_CHW (the layout the generator's map heads convolve): attributes written [b, c, h, w] by the kernel and
    the gradient taken from a [b, c, h, w] cotangent are, bit for bit, the reference layout's permuted — for the
    LDS-tiled and the global-key forward, three (register path) and five attribute channels, both precisions, and with
    a triangle large enough for the workgroup-cooperative gradient path."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth

    monkeypatch.setenv("SR_RASTER_TILED", tiled)
    dt = torch.float32 if dtype == "f32" else torch.float64
    v0, tri = synth.uv_ellipsoid(20, 18)
    vh = synth.random_poses(v0, 3, seed=19)
    nv = v0.shape[0]
    # one screen-filling triangle behind the mesh (the `big` list)
    vh = np.concatenate([vh, np.tile(np.array([[[-0.9, -0.9, -0.99], [0.9, -0.9, -0.99], [0.0, 0.9, -0.99]]], vh.dtype),
                                     (3, 1, 1))], 1)
    tri = np.concatenate([tri, [[nv, nv + 1, nv + 2]]], 0)
    tex = synth.det_normal((3, nv + 3, tex_c), 79)
    go = synth.det_normal((3, 64, 64, tex_c), 80)
    res = {}
    for chw in (False, True):
        v, t = T(vh).to(dt).requires_grad_(), T(tex).to(dt).requires_grad_()
        out = op.rasterize(v, t, T(tri), 64, channel_major=chw)
        g = T(go).to(dt)
        grads = torch.autograd.grad(out, [v, t], g.permute(0, 3, 1, 2).contiguous() if chw else g)
        res[chw] = (out.detach(), grads)
    assert res[True][0].shape == (3, tex_c, 64, 64) and res[True][0].is_contiguous()
    assert torch.equal(res[True][0], res[False][0].permute(0, 3, 1, 2))
    assert torch.equal(res[True][1][0], res[False][1][0]) and torch.equal(res[True][1][1], res[False][1][1])
    assert float(res[True][1][0].abs().max()) > 0


def test_rasterize_gradients_misc():
    """No-channel attributes, per-sample topology [b, nf, 3], only one of the two gradients requested, a
    triangle with a repeated vertex id and out-of-range ids (skipped like the reference does)."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth

    v0, tri = synth.uv_ellipsoid(12, 10)
    vh = synth.random_poses(v0, 2, seed=41)
    nv = v0.shape[0]
    tri = np.concatenate([tri, [[0, 0, 5]], [[1, 2, nv]], [[-1, 2, 3]]], 0)
    tri_b = np.stack([tri, tri[::-1].copy()], 0)
    tex1 = synth.det_normal((2, nv), 42)
    go = synth.det_normal((2, 48, 48), 43)
    v, tx = T(vh).requires_grad_(), T(tex1).requires_grad_()
    out = op.rasterize(v, tx, T(tri_b), 48)
    assert out.shape == (2, 48, 48)
    gv, gt = torch.autograd.grad(out, [v, tx], T(go))
    wv, wt = _oracle_grads(vh, tex1, tri_b, go, 48)
    assert np.abs(gv.cpu().numpy() - wv).max() <= 5e-5 * np.abs(wv).max()
    assert np.abs(gt.cpu().numpy() - wt).max() <= 5e-6 * np.abs(wt).max()
    v2 = T(vh).requires_grad_()
    (gv2,) = torch.autograd.grad(op.rasterize(v2, T(tex1), T(tri_b), 48), [v2], T(go))
    assert torch.equal(gv2, gv)
    tx2 = T(tex1).requires_grad_()
    (gt2,) = torch.autograd.grad(op.rasterize(T(vh), tx2, T(tri_b), 48), [tx2], T(go))
    assert torch.equal(gt2, gt)


def test_rasterize_rejects_bad_inputs():
    R = importlib.import_module("stylerenderer_amd.op.rasterize")

    v = torch.zeros(1, 3, 3, device=DEV)
    with pytest.raises(RuntimeError):
        R.forward(v, torch.zeros(1, 3, dtype=torch.int32, device=DEV), 4, 4)
    with pytest.raises(RuntimeError):
        R.forward(v.half(), torch.zeros(1, 3, dtype=torch.int64, device=DEV), 4, 4)
    with pytest.raises(RuntimeError):                     # both tensors on the same device (op/rasterize.cpp:122)
        R.forward(v, torch.zeros(1, 3, dtype=torch.int64), 4, 4)
    # CPU tensors are served by the host loops, like the reference's extension (op/rasterize.cpp:126-150)
    ic, cc = R.forward(torch.zeros(1, 3, 3), torch.zeros(1, 3, dtype=torch.int64), 4, 4)
    assert ic.device.type == "cpu" and not ic.any() and not cc.any()
    # empty triangle list: all background
    idx, coeff = R.forward(v, torch.zeros(0, 3, dtype=torch.int64, device=DEV), 4, 4)
    assert not idx.any() and not coeff.any()


# ------------------------------------------------------------------------------- fused element-wise
@pytest.mark.parametrize("shape,shared", [((2, 6, 8, 8), True), ((3, 5, 16, 12), False), ((2, 4, 4, 4), True),
                                          ((2, 3, 64, 68), False)])
def test_noise_bias_act_matches_two_step_path(shape, shared):
    """One fused pass == NoiseInjection followed by fused_leaky_relu, bit for bit, incl. all
    gradients and the double backward."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth
    from stylerenderer_amd.op.fused_elem import noise_bias_act

    b, c, h, w = shape
    x = T(synth.det_normal(shape, 81))
    noise = T(synth.det_normal((1 if shared else b, 1, h, w), 82))
    nw = T(synth.det_normal((1,), 83))
    bias = T(synth.det_normal((c,), 84))
    gy = T(synth.det_normal(shape, 85))

    def run(fused):
        xs, nws, bs = (t.clone().requires_grad_() for t in (x, nw, bias))
        if fused:
            y = noise_bias_act(xs, noise, nws, bs)
        else:
            y = op.fused_leaky_relu(xs + nws * noise, bs)
        g = gy.clone().requires_grad_()
        gx, gnw, gb = torch.autograd.grad(y, [xs, nws, bs], g, create_graph=True)
        probe = (gx * x).sum() + (gb * bias).sum() + (gnw * nw).sum()
        (gg,) = torch.autograd.grad(probe, g)
        return y, gx, gnw, gb, gg

    a, r = run(True), run(False)
    assert torch.equal(a[0], r[0]) and torch.equal(a[1], r[1])
    for i in (2, 3):
        assert torch.allclose(a[i], r[i], rtol=2e-5, atol=1e-5)
    assert torch.allclose(a[4], r[4], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hw", [(20, 24), (17, 17), (65, 65)])      # aligned rows / (2^k+1)^2 unaligned rows
def test_rowdot_matches_torch(hw):
    from stylerenderer_amd import synth
    from stylerenderer_amd.op.fused_elem import rowdot

    a = T(synth.det_normal((3, 7) + hw, 91)).requires_grad_()
    b = T(synth.det_normal((3, 7) + hw, 92)).requires_grad_()
    s = T(synth.det_normal((3, 7), 93)).requires_grad_()
    dots, out = rowdot(a, b, s)
    want = (a.double() * b.double()).sum((2, 3))
    assert torch.allclose(dots.double(), want, rtol=1e-5, atol=1e-5)
    assert torch.equal(out, b * s[:, :, None, None])
    d2 = rowdot(a, b)
    assert torch.equal(d2, dots)
    ga, gb, gs = torch.autograd.grad(dots.sum() + (out * a).sum(), [a, b, s])
    ra, rb, rs = torch.autograd.grad((a * b).sum() + (b * s[:, :, None, None] * a).sum(), [a, b, s])
    assert torch.allclose(ga, ra, rtol=1e-5, atol=1e-6) and torch.allclose(gb, rb, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gs, rs, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape,shared_noise", [((2, 5, 33, 33), False), ((3, 4, 9, 17), True), ((1, 6, 65, 65), False)])
def test_blur_noise_bias_act_equals_the_two_operators(shape, shared_noise):
    """Fused blur + noise/bias/LeakyReLU (k_fir4_tile<true>) == upfirdn2d followed by noise_bias_act, bit for
    bit (same tap order, same operation order), with identical gradients including second order."""
    from stylerenderer_amd import synth
    from stylerenderer_amd.op.fused_elem import blur_noise_bias_act, noise_bias_act
    from stylerenderer_amd.op.upfirdn2d import upfirdn2d

    n, c, h, w = shape
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    kernel = (k1[None, :] * k1[:, None] / 64.0 * 4.0).to(DEV)
    pad = (1, 1)
    oh, ow = h - 1, w - 1
    x = T(synth.det_normal(shape, 301)).requires_grad_()
    noise = T(synth.det_normal((1 if shared_noise else n, 1, oh, ow), 302))
    nw = T(synth.det_normal((1,), 303)).requires_grad_()
    bias = T(synth.det_normal((c,), 304)).requires_grad_()
    ref_in = [t.detach().clone().requires_grad_() for t in (x, nw, bias)]
    got = blur_noise_bias_act(x, kernel, pad, noise, nw, bias)
    blurred = upfirdn2d(ref_in[0], kernel, pad=pad)
    want = noise_bias_act(blurred, noise, ref_in[1], ref_in[2]) if (oh * ow) % 4 == 0 else None
    if want is None:          # noise_bias_act needs inner % 4 == 0 for its fused kernel: plain formula instead
        v = blurred + ref_in[1] * noise + ref_in[2][None, :, None, None]
        want = torch.where(v > 0, v, v * 0.2) * (2 ** 0.5)
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    else:
        assert torch.equal(got, want)
    proj = T(synth.det_normal(tuple(got.shape), 305))
    g1 = torch.autograd.grad((got * proj).sum(), [x, nw, bias], create_graph=True)
    g2 = torch.autograd.grad((want * proj).sum(), ref_in, create_graph=True)
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    h1 = torch.autograd.grad((g1[0] * g1[0]).sum(), nw)[0]
    h2 = torch.autograd.grad((g2[0] * g2[0]).sum(), ref_in[1])[0]
    assert torch.allclose(h1, h2, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape,shared,cmap", [((2, 6, 8, 8), True, 2), ((3, 5, 16, 12), False, 4), ((4, 128, 32, 32), False, 4)])
def test_noise_bias_act_affine_matches_unfused_path(shape, shared, cmap):
    """StyledMapConv tail (reference model.py:49-54) in one pass == mul, add, NoiseInjection, fused_leaky_relu:
    forward bit for bit; gradients (x, both map channels, noise strength, bias) to fp32 round-off of the channel
    sums; the recorded backward (path-length regulariser) through the same node."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth
    from stylerenderer_amd.op.fused_elem import noise_bias_act_affine

    b, c, h, w = shape
    x0 = T(synth.det_normal(shape, 91))
    m0 = T(synth.det_normal((b, cmap, h, w), 92))
    noise = T(synth.det_normal((1 if shared else b, 1, h, w), 93))
    nw0, b0 = T(synth.det_normal((1,), 94)), T(synth.det_normal((c,), 95))
    gy = T(synth.det_normal(shape, 96))
    lo = cmap - 2

    def run(fused):
        x, m, nw, bb = (t.clone().requires_grad_() for t in (x0, m0, nw0, b0))
        sm = m[:, lo:lo + 2]
        if fused:
            y = noise_bias_act_affine(x, sm, noise, nw, bb)
        else:
            y = op.fused_leaky_relu(x * sm[:, :1] + sm[:, 1:2] + nw * noise, bb)
        return (x, m, nw, bb), y

    (ia, ya), (ib, yb) = run(True), run(False)
    assert bits_equal(ya.detach().cpu().numpy(), yb.detach().cpu().numpy())
    ga = torch.autograd.grad(ya, ia, gy)
    gb = torch.autograd.grad(yb, ib, gy)
    for u, v in zip(ga, gb):
        assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 1e-6
    # double backward: gradient of |d y / d x|^2-style functional w.r.t. everything
    def second(inputs, y):
        (g1, gm) = torch.autograd.grad(y, [inputs[0], inputs[1]], gy, create_graph=True)
        return torch.autograd.grad((g1 * g1).sum() + (gm * gm).sum(), inputs, allow_unused=True)

    (ia, ya), (ib, yb) = run(True), run(False)
    for u, v in zip(second(ia, ya), second(ib, yb)):
        if v is None:
            assert u is None or float(u.abs().max()) == 0
            continue
        assert float((u - v).abs().max()) <= 1e-4 * float(v.abs().max()) + 1e-6


def test_empty_and_degenerate_inputs_on_device():
    """Zero-size tensors and degenerate geometry through every operator of the boundary: no launch, no crash,
    shapes and gradients as torch would produce them."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd.op.fused_elem import noise_bias_act
    R = importlib.import_module("stylerenderer_amd.op.rasterize")

    bias = torch.zeros(4, device=DEV, requires_grad=True)
    x = torch.zeros(0, 4, 3, 3, device=DEV, requires_grad=True)
    y = op.fused_leaky_relu(x, bias)
    assert y.shape == (0, 4, 3, 3)
    y.sum().backward()
    assert bias.grad is not None and float(bias.grad.abs().max()) == 0.0
    k = torch.ones(4, 4, device=DEV) / 16
    assert op.upfirdn2d(torch.zeros(0, 3, 8, 8, device=DEV), k, pad=(1, 1)).shape == (0, 3, 7, 7)
    assert op.upfirdn2d(torch.zeros(2, 0, 8, 8, device=DEV), k, up=2, pad=(2, 1)).shape == (2, 0, 16, 16)
    one = op.upfirdn2d(torch.ones(1, 1, 1, 1, device=DEV), k, pad=(2, 1))            # 1x1 image, kernel larger than it
    assert one.shape == (1, 1, 1, 1) and abs(float(one) - 1 / 16) < 1e-7
    assert noise_bias_act(torch.zeros(0, 4, 8, 8, device=DEV), None, None, bias.detach()).shape == (0, 4, 8, 8)
    # rasterizer: empty batch, no triangles, a single vertex referenced three times, 1x1 target
    v = torch.zeros(2, 3, 3, device=DEV)
    tri0 = torch.zeros(0, 3, dtype=torch.int64, device=DEV)
    out = op.rasterize(v.clone().requires_grad_(), torch.ones(2, 3, 2, device=DEV).requires_grad_(), tri0, 4)
    assert out.shape == (2, 4, 4, 2) and not out.any()
    gv, gt = torch.autograd.grad(out.sum(), [n for n in (out.grad_fn.next_functions[0][0].variable,
                                                         out.grad_fn.next_functions[1][0].variable)])
    assert not gv.any() and not gt.any()
    idx, coeff = R.forward(torch.zeros(0, 3, 3, device=DEV), tri0, 4, 4)
    assert idx.shape == (0, 4, 4, 3) and coeff.shape == (0, 4, 4, 3)
    same = torch.zeros(1, 3, dtype=torch.int64, device=DEV)                          # ids (0, 0, 0): a point
    idx, coeff = R.forward(v, same, 1, 1)
    assert idx.shape == (2, 1, 1, 3) and torch.isfinite(coeff).all()


@pytest.mark.parametrize("up,down,pad", [(1, 2, (1, 1)), (1, 2, (2, 2)), (2, 1, (2, 1)), (2, 1, (1, 1)), (1, 2, (-1, 3))])
@pytest.mark.parametrize("shape", [(2, 3, 70, 45), (1, 2, 33, 129), (1, 1, 5, 3)])
def test_upfirdn2d_resampling_tiles_vs_oracle(up, down, pad, shape):
    """The LDS-tiled 4x4 resampling kernels (k_fir4_resample: down = 2 and up = 2) over several tiles, odd sizes,
    crops: bitwise equal to the numpy oracle, and gradients bitwise equal to the oracle applied to the adjoint
    parameters (the other resampling kernel)."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth

    x = synth.det_normal(shape, 201)
    k = ops_np.make_blur_kernel((1, 3, 3, 1), float(up * up))
    xt = T(x).requires_grad_()
    y = op.upfirdn2d(xt, T(k), up=up, down=down, pad=pad)
    want = ops_np.upfirdn2d(x, k, up, down, pad)
    assert bits_equal(y.detach().cpu().numpy(), want)
    gy = synth.det_normal(tuple(y.shape), 202)
    (gx,) = torch.autograd.grad(y, xt, T(gy))
    in_h, in_w = shape[2], shape[3]
    gx0, gx1 = 4 - pad[0] - 1, in_w * up - y.shape[3] * down + pad[0] - up + 1
    gy0, gy1 = 4 - pad[0] - 1, in_h * up - y.shape[2] * down + pad[0] - up + 1
    want_gx = ops_np.upfirdn2d_full(gy, k[::-1, ::-1].copy(), down, down, up, up, gx0, gx1, gy0, gy1)
    assert bits_equal(gx.cpu().numpy(), want_gx)


def test_torgb_fused_bias_and_skip_add_equal_the_separate_operators():
    """ToRGB on device tensors = streaming 1x1 kernel with the bias + up-sampling kernel with the skip addition
    (two launches) against conv, + bias, upsample, + (reference model.py:63-69): outputs bit for bit, gradients and
    the double backward to round-off."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import model, synth

    rgb = model.ToRGB(16, 32).to(DEV)
    synth.fill_state_dict(rgb.state_dict(), salt=71)
    x0, s0 = T(synth.det_normal((2, 16, 12, 20), 72)), T(synth.det_normal((2, 32), 73))
    k0 = T(synth.det_normal((2, 3, 6, 10), 74))
    proj = T(synth.det_normal((2, 3, 12, 20), 75))

    def run(fused):
        x, s, k = (t.clone().requires_grad_() for t in (x0, s0, k0))
        if fused:
            y = rgb(x, s, k)
        else:
            y = rgb.conv(x, s) + rgb.bias
            y = y + op.upfirdn2d(k, rgb.upsample.kernel, up=2, pad=rgb.upsample.pad)
        params = [x, s, k, rgb.bias, rgb.conv.weight]
        g1 = torch.autograd.grad((y * proj).sum(), params, create_graph=True)
        g2 = torch.autograd.grad((g1[0] * g1[0]).sum() + (g1[2] * g1[2]).sum(), [s, rgb.conv.weight], allow_unused=True)
        return y.detach(), [g.detach() for g in g1], g2

    ya, ga, ha = run(True)
    yb, gb, hb = run(False)
    assert torch.equal(ya, yb)
    for u, v in zip(ga, gb):
        assert float((u - v).abs().max()) <= 1e-6 * float(v.abs().max()) + 1e-9
    for u, v in zip(ha, hb):
        assert (u is None) == (v is None)
        if v is not None:
            assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-9


@pytest.mark.parametrize("persp", [False, True])
def test_rasterize_tiled_path_list_overflow_wide_boxes_and_tile_borders(persp, monkeypatch):
    """The LDS-tiled forward (k_tile_bin / k_tile_raster) on inputs built to leave its common path: > 2 048 small
    triangles inside ONE 32x32 tile (list capacity overflow -> the sample's wide list), boxes that straddle two, three
    and four tiles, boxes larger than 64 pixels (wide list), equal depths (ties -> lowest id) and an image whose edge
    is not a multiple of the tile (80).  Bit-exact against the C oracle, like every other rasterizer test."""
    R = importlib.import_module("stylerenderer_amd.op.rasterize")
    monkeypatch.setenv("SR_RASTER_TILED", "1")
    rng = np.random.RandomState(11)
    res = 80
    tris = []

    def add(cx, cy, size, z, n=1):
        for _ in range(n):
            a = rng.rand() * 2 * np.pi
            pts = [(cx + size * np.cos(a + k * 2.1), cy + size * np.sin(a + k * 2.1)) for k in range(3)]
            tris.append([(x, y, z + 0.01 * rng.randn()) for x, y in pts])

    for _ in range(6000):                                   # one tile (pixels 32..63 x 0..31): ~3 000 accepted > capacity 2 048
        add(32 + 32 * rng.rand(), 32 * rng.rand(), 0.6 + 1.5 * rng.rand(), -2.0 - rng.rand())
    for _ in range(300):                                    # straddle the tile borders at 32 and 64
        add(rng.choice([31.5, 63.5]) + rng.randn(), rng.choice([31.5, 63.5]) + rng.randn(), 1.0 + 2 * rng.rand(), -2.5)
    for _ in range(40):                                     # thin strips: three tiles in a row
        y = 80 * rng.rand()
        tris.append([(20.0, y, -3.0), (75.0, y + 0.6, -3.0), (47.0, y + 1.2, -3.0)])
    for _ in range(12):                                     # large boxes
        add(80 * rng.rand(), 80 * rng.rand(), 10 + 25 * rng.rand(), -3.5)
    for _ in range(60):                                     # exact ties: identical triangles, different ids
        tris.append(tris[int(rng.randint(0, 3000))])
    pix = np.asarray(tris, np.float64)                      # [nf, 3, (px, py, z)]
    ndc = np.empty_like(pix)
    ndc[..., 0] = (pix[..., 0] + 0.5) * 2 / res - 1
    ndc[..., 1] = 1 - (pix[..., 1] + 0.5) * 2 / res
    ndc[..., 2] = pix[..., 2]
    if persp:
        ndc[..., 0] *= -ndc[..., 2]
        ndc[..., 1] *= -ndc[..., 2]
    v1 = ndc.reshape(-1, 3).astype(np.float32)
    nf = pix.shape[0]
    tri = np.arange(3 * nf, dtype=np.int64).reshape(nf, 3)
    flip = rng.rand(nf) < 0.5                               # both windings: half are culled
    tri[flip] = tri[flip][:, ::-1]
    v = np.stack([v1, v1 * np.float32(0.97)], 0)
    want_i, want_c, want_z = raster.forward_buffers(v, tri, res, res, persp, 1e-6)
    idx, coeff, zbuf = R.forward_with_depth(T(v), T(tri), res, res, persp, 1e-6)
    assert np.array_equal(idx.cpu().numpy(), want_i)
    assert bits_equal(coeff.cpu().numpy(), want_c)
    assert bits_equal(zbuf.cpu().numpy(), want_z)
    covered = int((want_i != 0).any(-1).sum())
    assert covered > 2000


def test_frozen_bias_and_noise_strength_skip_their_reductions_same_input_gradients():
    """With frozen parameters (inversion, sampling, the discriminator inside the generator's phase) the fused
    activation nodes skip the bias / noise-strength reductions: the gradients with respect to the activations — first
    and second order — are the same bits, the parameter gradients are simply absent."""
    from stylerenderer_amd.op import fused_leaky_relu
    from stylerenderer_amd.op.fused_elem import noise_bias_act, noise_bias_act_affine

    g = torch.Generator().manual_seed(21)
    x0 = torch.randn(3, 8, 16, 16, generator=g).to(DEV)
    noise = torch.randn(3, 1, 16, 16, generator=g).to(DEV)
    smap = torch.randn(3, 2, 16, 16, generator=g).to(DEV)
    gy = torch.randn(3, 8, 16, 16, generator=g).to(DEV)
    b0, nw0 = torch.randn(8, generator=g).to(DEV), torch.randn(1, generator=g).to(DEV)

    def run(kind, frozen):
        x = x0.clone().requires_grad_()
        bias = b0.clone().requires_grad_(not frozen)
        nw = nw0.clone().requires_grad_(not frozen)
        if kind == "act":
            y = fused_leaky_relu(x, bias)
        elif kind == "nba":
            y = noise_bias_act(x, noise, nw, bias)
        else:
            y = noise_bias_act_affine(x, smap, noise, nw, bias)
        (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
        (ggy,) = torch.autograd.grad(gx.square().sum(), x, allow_unused=True) if kind == "aff" else (None,)
        y2 = fused_leaky_relu(x, bias) if kind == "act" else (noise_bias_act(x, noise, nw, bias) if kind == "nba"
                                                               else noise_bias_act_affine(x, smap, noise, nw, bias))
        y2.backward(gy)
        return (y.detach(), gx.detach(), x.grad.clone(), ggy, bias.grad, nw.grad)

    for kind in ("act", "nba", "aff"):
        a, f = run(kind, False), run(kind, True)
        for u, v in zip(a[:3], f[:3]):
            assert torch.equal(u, v), kind
        if a[3] is not None:
            assert torch.equal(a[3], f[3]), kind
        assert a[4] is not None and f[4] is None, kind
        if kind != "act":
            assert a[5] is not None and f[5] is None, kind


@pytest.mark.parametrize("shared", [True, False])
def test_affine3_pose_kernel_vs_matmul(shared):
    """utils_3d.affine3 (sr_affine3_fwd / _bwd): v @ M + t and its three gradients against the library matmul it
    replaces (fp32 round-off; the 3x3 / translation gradients are fixed-order sums: two launches agree bit for bit)."""
    from stylerenderer_amd import synth, utils_3d

    b, nv = 3, 24770
    v = T(synth.det_normal((1 if shared else b, nv, 3), 11)).requires_grad_()
    m = T(synth.det_normal((b, 3, 3), 12)).requires_grad_()
    t = T(synth.det_normal((b, 3), 13)).requires_grad_()
    go = T(synth.det_normal((b, nv, 3), 14))
    out = utils_3d.affine3(v, m, t)
    want = torch.matmul(v, m) + t.view(b, 1, 3)
    assert out.shape == want.shape and rel_err(out.detach().cpu().numpy(), want.detach().cpu().numpy()) < 1e-6
    got = torch.autograd.grad(out, [v, m, t], go)
    ref = torch.autograd.grad(want, [v, m, t], go)
    for a, r, tol in zip(got, ref, (2e-6, 2e-5, 2e-5)):
        assert a.shape == r.shape and rel_err(a.cpu().numpy(), r.cpu().numpy()) < tol
    again = torch.autograd.grad(utils_3d.affine3(v, m, t), [m, t], go)
    assert torch.equal(again[0], got[1]) and torch.equal(again[1], got[2])
    out2 = utils_3d.affine3(v, m)                                  # no translation (normals)
    assert rel_err(out2.detach().cpu().numpy(), torch.matmul(v, m).detach().cpu().numpy()) < 1e-6


def test_pose_matrices_kernel_vs_tensor_algebra():
    """utils_3d.pose_matrices (sr_pose_fwd / _bwd, one lane each) against euler_mat + exp, values and gradient."""
    from stylerenderer_amd import utils_3d

    pose = torch.tensor([0.31, -0.12, 0.07, 0.2, -0.1, 0.05, 0.15], device=DEV, requires_grad=True)
    lin, rot = utils_3d.pose_matrices(pose)
    ref_rot = utils_3d.euler_mat(pose[:3].view(1, 3), "yxz")
    ref_lin = torch.exp(pose[6]) * ref_rot
    assert lin.shape == (1, 3, 3) and rot.shape == (1, 3, 3)
    assert torch.allclose(rot, ref_rot, atol=2e-7) and torch.allclose(lin, ref_lin, atol=3e-7)
    gl = torch.randn(1, 3, 3, device=DEV)
    gr = torch.randn(1, 3, 3, device=DEV)
    (got,) = torch.autograd.grad((lin * gl).sum() + (rot * gr).sum(), pose)
    (want,) = torch.autograd.grad((ref_lin * gl).sum() + (ref_rot * gr).sum(), pose)
    assert torch.allclose(got, want, atol=3e-6), (got, want)
    (only_rot,) = torch.autograd.grad((utils_3d.pose_matrices(pose)[1] * gr).sum(), pose)
    (want_rot,) = torch.autograd.grad((utils_3d.euler_mat(pose[:3].view(1, 3), "yxz") * gr).sum(), pose)
    assert torch.allclose(only_rot, want_rot, atol=3e-6)


@pytest.mark.gpu
def test_random_pose_batch_kernel_vs_tensor_algebra():
    """utils_3d.random_apply_pose3D on the device (sr_pose_batch_fwd + sr_affine3_fwd: two launches, capturable) against
    the reference's tensor algebra (utils_3d.py:360-376) on the same draw: z = randn * sigma from the same seed."""
    from stylerenderer_amd import utils_3d

    sig = torch.tensor([.5, .1, .05, .1, .1, .1, .15], device=DEV)
    for batch, nv in ((1, 17), (4, 1000), (67, 33)):
        v = torch.randn(batch, nv, 3, device=DEV)
        torch.manual_seed(11 + batch)
        got = utils_3d.random_apply_pose3D(p=sig, v=v)
        torch.manual_seed(11 + batch)
        z = torch.randn(batch, 7, device=DEV) * sig
        T = torch.exp(z[:, -1]).view(-1, 1, 1) * utils_3d.euler_mat(z[:, :3], "yxz")
        want = torch.matmul(v, T) + z[:, 3:6].view(-1, 1, 3)
        assert got.shape == want.shape and torch.allclose(got, want, atol=3e-6), (got - want).abs().max()
    # the sampling is capturable: no host read between the draw and the posed vertices
    g = torch.cuda.CUDAGraph()
    v = torch.randn(4, 100, 3, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        utils_3d.random_apply_pose3D(p=sig, v=v)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = utils_3d.random_apply_pose3D(p=sig, v=v)
    g.replay()
    first = out.clone()
    g.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and not torch.equal(first, out)       # a fresh draw per replay
