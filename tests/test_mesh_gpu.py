"""GPU: vertex-normal gather kernel (csrc/mesh.hip through stylerenderer_amd.utils_3d) against the
reference's golden output, against the same tensor algebra in float64, run-to-run bit identity,
gradients, and the face-sized mesh of BASELINE config[3] feeding the rasterizer."""
import numpy as np
import pytest
import torch

from stylerenderer_amd import synth, utils_3d
from util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = torch.from_numpy


def test_vertex_normals_match_reference_golden(golden):
    g = golden("mesh_frontend")
    v = T(g["v"]).to(DEV).requires_grad_(True)
    tri = T(g["tri"].astype(np.int64)).to(DEV)
    n = utils_3d.mesh_point_normal(v, tri)
    # same association as the reference's three scatters: agreement to the last bits
    assert rel_err(n.detach().cpu().numpy(), g["normals"]) < 5e-7
    proj = T(synth.det_normal(tuple(n.shape), 71)).to(DEV)
    (gv,) = torch.autograd.grad((n * proj).sum(), v)
    assert rel_err(gv.cpu().numpy(), g["grad_v"]) < 1e-5


def test_vertex_normals_face_sized_mesh_deterministic_and_feed_rasterizer():
    from stylerenderer_amd import op

    v0, tri = synth.face_sized_mesh()
    v = T(synth.random_poses(v0, 4, seed=5)).to(DEV)
    trit = T(tri.astype(np.int64)).to(DEV)
    a = utils_3d.mesh_point_normal(v, trit)
    b = utils_3d.mesh_point_normal(v, trit)
    assert torch.equal(a, b)                                   # gather in a fixed order: bit identical
    want = utils_3d._normals_composite(v.double().cpu(), trit.cpu())
    assert rel_err(a.cpu().numpy(), want.numpy()) < 2e-5       # cancellation in near-degenerate fans
    lens = a.norm(dim=2)
    assert float((lens - 1).abs().max()) < 1e-5
    img = op.rasterize(v, a, trit, 64, 64)
    assert tuple(img.shape) == (4, 64, 64, 3) and torch.isfinite(img).all()
    assert float(img.abs().sum()) > 0
