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


def test_full_size_training_iteration_256_on_hip_path():
    """BASELINE config[2] at size: GeneratorWithMap(256) + Discriminator(256), 4 images per GPU, 3DMM-size
    mesh sampled per step, iteration 0 (R1 + path-length regulariser on batch 2, i.e. the double backward
    through every HIP operator at 256x256) and iteration 1 (plain).  Checks finiteness, that every trainable
    parameter of both networks moved, and that the EMA follows."""
    dev = torch.device("cuda")
    tr = train.Trainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0)
    data = train.SyntheticImages(8, 256, dev)
    faces = train.SyntheticFaceSource(dev, seed=0)
    g0 = {n: p.detach().clone() for n, p in tr.generator.named_parameters()}
    d0 = {n: p.detach().clone() for n, p in tr.discriminator.named_parameters()}
    e0 = {n: p.detach().clone() for n, p in tr.g_ema.named_parameters()}
    logs = [tr.step(data.batch(4), faces=faces) for _ in range(2)]
    assert {"d", "g", "r1", "path", "path_length", "real_score", "fake_score", "mean_path"} <= set(logs[0])
    assert "r1" not in logs[1] and "path" not in logs[1]
    for log in logs:
        assert all(np.isfinite(v) for v in log.values()), log
    assert logs[0]["path_length"] > 0
    stuck = [n for n, p in tr.generator.named_parameters() if n not in tr.frozen and torch.equal(p, g0[n])]
    assert not stuck, stuck
    assert all(torch.equal(p, g0[n]) for n, p in tr.generator.named_parameters() if n in tr.frozen)
    stuck = [n for n, p in tr.discriminator.named_parameters() if torch.equal(p, d0[n])]
    assert not stuck, stuck
    assert any(not torch.equal(p, e0[n]) for n, p in tr.g_ema.named_parameters())
    for n, p in list(tr.generator.named_parameters()) + list(tr.discriminator.named_parameters()):
        assert torch.isfinite(p).all(), n
