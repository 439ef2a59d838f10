"""CPU tensors through the PRODUCT's host path (libstylerenderer_hip.so sr_rasterize_*_cpu_*: the reference's
extension also serves CPU tensors, op/rasterize.cpp:126-150) against the golden vectors generated from the
reference's own op/rasterize.cpp and against the C oracle — bitwise for index / coeff / z-buffer / dcoeff."""
import importlib

import numpy as np
import pytest
import torch

import raster
from util import bits_equal

R = importlib.import_module("stylerenderer_amd.op.rasterize")
T = torch.from_numpy


@pytest.mark.parametrize("name", ["raster_ellipsoid_32", "raster_ellipsoid_64", "raster_perspective_32",
                                  "raster_adversarial_16"])
def test_host_path_bitexact_vs_golden(golden, name):
    g = golden(name)
    persp = "perspective" in name
    res = g["index"].shape[1]
    v, tri = T(g["v"]), T(g["tri"].astype(np.int64))
    idx, coeff, zbuf = R.forward_with_depth(v, tri, res, res, persp, 1e-6)
    assert np.array_equal(idx.numpy(), g["index"].astype(np.int64))
    assert bits_equal(coeff.numpy(), g["coeff"]) and bits_equal(zbuf.numpy(), g["zbuf"])
    assert bits_equal(R.backward(v, idx, persp, 1e-6).numpy(), g["dcoeff"])


def test_host_path_misc_topologies(golden):
    g = golden("raster_misc")
    idx, coeff, zbuf = R.forward_with_depth(T(g["v"]), T(g["tri_b"].astype(np.int64)), 16, 16, False, 1e-6)
    assert np.array_equal(idx.numpy(), g["index_b"].astype(np.int64))
    assert bits_equal(coeff.numpy(), g["coeff_b"]) and bits_equal(zbuf.numpy(), g["zbuf_b"])
    idx, coeff = R.forward(T(g["v_bf"]), T(g["tri_bf"].astype(np.int64)), 32, 32, False, 1e-6)
    assert np.array_equal(idx.numpy(), g["index_bf"].astype(np.int64)) and bits_equal(coeff.numpy(), g["coeff_bf"])


def test_host_path_known_answer_fp64_and_gradcheck(golden):
    """The reference's __main__ test (op/rasterize.py:83-107) on CPU tensors."""
    import stylerenderer_amd.op as op

    g = golden("raster_kat")
    v, t, f = T(g["v"]).requires_grad_(), T(g["tex"]).requires_grad_(), T(g["f"])
    o = op.rasterize(v, t, f, 5)
    assert np.abs(o.detach().numpy() - g["out"]).max() < 1e-15
    gv, gt = torch.autograd.grad(o, [v, t], T(g["grad_out"]))
    # the reference scatters through a float32 sparse matrix (op/rasterize.py:63-64): ITS gradients carry fp32
    # round-off even for float64 inputs
    assert np.allclose(gv.numpy(), g["grad_v"], rtol=2e-6, atol=1e-7)
    assert np.allclose(gt.numpy(), g["grad_tex"], rtol=2e-6, atol=1e-7)
    cat = torch.cat((v.detach(), t.detach()), -1).requires_grad_()
    assert torch.autograd.gradcheck(lambda x: op.rasterize(x[:, :, :3], x[:, :, 3:], f, 5), cat, eps=1e-6, atol=1e-6)


@pytest.mark.parametrize("res", [32, 64])
def test_host_path_interp_and_grads(golden, res):
    import stylerenderer_amd.op as op

    g = golden("raster_ellipsoid_%d" % res)
    v, tex, tri = T(g["v"]).requires_grad_(), T(g["tex"]).requires_grad_(), T(g["tri"].astype(np.int64))
    out = op.rasterize(v, tex, tri, res)
    assert bits_equal(out.detach().numpy(), raster.rasterize(g["v"], g["tex"], g["tri"].astype(np.int64), res))
    gv, gt = torch.autograd.grad(out, [v, tex], T(g["grad_out"]))
    assert np.abs(gv.numpy() - g["grad_v"]).max() <= 2e-5 * np.abs(g["grad_v"]).max()
    assert np.abs(gt.numpy() - g["grad_tex"]).max() <= 2e-6 * np.abs(g["grad_tex"]).max()


def test_generator_with_map_runs_on_cpu():
    """BASELINE config[0]-style plumbing for the mesh-conditioned generator: no GPU anywhere."""
    from stylerenderer_amd import model, synth

    g = model.GeneratorWithMap(8, 32, 2)
    v0, tri = synth.uv_ellipsoid(8, 8)
    v = synth.random_poses(v0, 2, seed=1)
    mesh = (T(v), T(synth.vertex_normals(v, tri)), T(tri))
    img, _, maps = g([torch.randn(2, 32)], mesh, return_normals=True)
    assert img.shape == (2, 3, 8, 8) and len(maps) == 2 and torch.isfinite(img).all()
