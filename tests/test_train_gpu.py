"""GPU: one full training iteration (D, R1, G, path-length regulariser with double backward through
every HIP operator, EMA) for both generator flavours, on the HIP path."""
import numpy as np
import pytest
import torch

from stylerenderer_amd import train

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_mesh", [False, True])
def test_training_iteration_on_device(use_mesh):
    tr = train.Trainer(size=16, latent=32, n_mlp=2, use_mesh=use_mesh, device="cuda", seed=2)
    data = train.SyntheticImages(8, 16, "cuda")
    mesh = train.synthetic_mesh(4, "cuda", seed=1, face_sized=False) if use_mesh else None
    before = tr.generator.conv1.conv.weight.detach().clone()
    logs = [tr.step(data.batch(4), mesh) for _ in range(2)]
    assert {"d", "g", "r1", "path", "path_length"} <= set(logs[0])
    for log in logs:
        assert all(np.isfinite(v) for v in log.values()), log
    assert not torch.equal(before, tr.generator.conv1.conv.weight)
    assert float(tr.mean_path_length) > 0


def test_synthetic_face_source_feeds_training_step():
    """3DMM -> pose -> vertex normals on the device (reference train.py:246-251) into a training step."""
    faces = train.SyntheticFaceSource(torch.device("cuda"), shape_dim=6, expression_dim=4, seed=3, face_sized=False)
    vert, nrm, tri = faces.sample(4)
    assert tuple(vert.shape) == tuple(nrm.shape) and vert.shape[0] == 4 and vert.is_cuda
    assert float((nrm.norm(dim=2) - 1).abs().max()) < 1e-4
    tr = train.Trainer(size=16, latent=32, n_mlp=2, use_mesh=True, device="cuda", seed=2)
    data = train.SyntheticImages(8, 16, "cuda")
    log = tr.step(data.batch(4), faces.sample(4))
    assert all(np.isfinite(v) for v in log.values()), log
