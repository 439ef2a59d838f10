"""GPU: one full training iteration (D, R1, G, path-length regulariser with double backward through
every HIP operator, EMA) for both generator flavours, on the HIP path."""
import numpy as np
import pytest
import torch

from stylerenderer_amd import synth, train

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


@pytest.mark.parametrize("use_mesh", [False, True])
def test_graphed_training_iteration(use_mesh):
    """graph_train.GraphedTrainer: every phase (D, R1, G, path-length double backward) and both Adam steps
    replayed from hipGraphs.  Losses finite, parameters move, successive replays draw fresh latents / noise
    (graph-safe Philox), and the captured step stays close to the same bodies run eagerly from the same state."""
    from stylerenderer_amd import graph_train

    dev = torch.device("cuda")
    faces = train.SyntheticFaceSource(dev, shape_dim=6, expression_dim=4, seed=3, face_sized=False) if use_mesh else None
    kw = dict(size=16, latent=32, n_mlp=2, use_mesh=use_mesh, device=dev, seed=2, batch=4,
              mesh_vertices=(faces.model.dim[2] // 3 if use_mesh else None))
    tr = graph_train.GraphedTrainer(**kw)
    data = train.SyntheticImages(8, 16, dev)
    before = tr.generator.conv1.conv.weight.detach().clone()
    d_before = tr.discriminator.final_conv[0].weight.detach().clone()
    logs = [tr.step(data.batch(4), faces=faces) for _ in range(5)]
    assert tr.graphs and set(tr.graphs) == {"d", "r1", "g", "path", "d_opt", "g_opt", "ema"}
    assert {"d", "g", "r1", "path", "path_length", "mean_path"} <= set(logs[0])
    assert "r1" not in logs[1] and "path" not in logs[1] and "path" in logs[4]
    for log in logs:
        assert all(np.isfinite(v) for v in log.values()), log
    assert len({round(log["g"], 6) for log in logs}) == len(logs)          # fresh randomness per replay
    assert not torch.equal(before, tr.generator.conv1.conv.weight)
    assert not torch.equal(d_before, tr.discriminator.final_conv[0].weight)
    assert float(tr.mean_path_length) > 0
    # frozen ToRGB tail never moves
    ref = graph_train.GraphedTrainer(**kw)
    assert all(torch.equal(p, q) for (n, p), (_, q) in zip(tr.generator.named_parameters(),
                                                          ref.generator.named_parameters()) if n in tr.frozen)
    st = tr.state_dict()
    assert len(st["g_optim"]["param_groups"][0]["params"]) == sum(1 for _ in tr.generator.parameters())


def test_graphed_step_matches_eager_bodies():
    """Same trainer state, same device RNG state: the captured D phase and the eagerly executed D phase leave
    the same gradients (capture changes where kernels are launched from, not what they compute)."""
    from stylerenderer_amd import graph_train

    dev = torch.device("cuda")
    tr = graph_train.GraphedTrainer(size=16, latent=32, n_mlp=2, device=dev, seed=5, batch=4)
    data = train.SyntheticImages(8, 16, dev)
    tr.step(data.batch(4))
    tr.s_real.copy_(data.batch(4))
    state = torch.cuda.get_rng_state(dev)
    tr._phase_d()
    eager = tr.flat_d.clone()
    torch.cuda.set_rng_state(state, dev)
    tr.graphs["d"].replay()
    torch.cuda.synchronize()
    scale = float(eager.abs().max())
    assert scale > 0 and float((tr.flat_d - eager).abs().max()) <= 1e-5 * scale


def test_train_functions_on_hip_path_match_reference_definitions():
    """train.d_logistic_loss / d_r1_loss / g_nonsaturating_loss / g_path_regularize (two targets, lambda_ weights,
    double backward) with device tensors against tests/golden/train_step_s8.npz (reference train.py:100-134).
    Measured: activations 1e-6, R1 gradient samples 1e-4 (second-order through the 8x8 discriminator), path-length
    gradient samples 1.6e-6."""
    import os

    import test_train_parity_cpu as tp

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_step_s8.npz"))
    tp.run_all(gold, "cuda", 2e-5, 2e-4, 4e-4)


def test_flat_adam_matches_torch_adam_and_speaks_its_checkpoint_format():
    """optim.FlatAdam (one sr_adam_flat launch over flat buffers) against torch.optim.Adam on the same gradients,
    with the lazy-regularisation hyper-parameters of the G optimiser (beta1 = 0), incl. a state_dict round trip."""
    from stylerenderer_amd import optim as sr_optim

    dev = torch.device("cuda")
    shapes = [(7, 5, 3, 3), (130,), (1,), (64, 33)]
    ref = [torch.from_numpy(synth.det_normal(s, 11 + i)).to(dev).requires_grad_() for i, s in enumerate(shapes)]
    mine = [p.detach().clone().requires_grad_() for p in ref]
    offs, total = sr_optim.flat_layout(mine)
    flat_g = torch.zeros(total, device=dev)
    lr, betas = 0.002 * 0.8, (0.0, 0.99 ** 0.8)
    a = torch.optim.Adam(ref, lr=lr, betas=betas)
    b = sr_optim.FlatAdam(mine, flat_g, lr=lr, betas=betas)
    assert all(p.data_ptr() % 256 == 0 for p in mine)
    gviews = sr_optim.flat_views(flat_g, mine, offs)
    for step in range(5):
        if step == 3:                                   # checkpoint round trip through torch's format
            sd = b.state_dict()
            assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3
            b2 = sr_optim.FlatAdam([p.detach().clone().requires_grad_() for p in mine], torch.zeros_like(flat_g),
                                   lr=lr, betas=betas)
            b2.load_state_dict(sd)
            assert torch.equal(b2.m, b.m) and torch.equal(b2.v, b.v) and float(b2.step_t) == 3
            c = torch.optim.Adam([p.detach().clone().requires_grad_() for p in ref], lr=lr, betas=betas)
            c.load_state_dict(a.state_dict())           # the reverse direction: torch loads what torch wrote
        for i, (p, q) in enumerate(zip(ref, mine)):
            g = torch.from_numpy(synth.det_normal(shapes[i], 100 * step + i)).to(dev) * (0.1 + step)
            p.grad = g.clone()
            gviews[i].copy_(g)
        a.step()
        b.step()
        for p, q in zip(ref, mine):
            assert float((p - q).abs().max()) <= 2e-6 * float(p.abs().max()) + 1e-7, step


def test_checkpoint_moves_between_graphed_and_eager_trainers(tmp_path):
    """The reference's checkpoint dict written by the graph-replayed trainer (flat-buffer Adam) resumes in the eager
    trainer (torch.optim.Adam) and the other way round: weights, EMA, both optimisers' moments, iteration."""
    from stylerenderer_amd import checkpoint, graph_train

    dev = torch.device("cuda")
    kw = dict(size=16, latent=32, n_mlp=2, device=dev, seed=4)
    a = graph_train.GraphedTrainer(batch=4, **kw)
    data = train.SyntheticImages(8, 16, dev)
    for _ in range(3):
        a.step(data.batch(4))
    path = checkpoint.save_checkpoint(str(tmp_path / checkpoint.checkpoint_name(3)), a)
    ck = torch.load(path, weights_only=False)
    assert set(checkpoint.CKPT_KEYS) <= set(ck) and ck["iteration"] == 3
    n_all = sum(1 for _ in a.generator.parameters())
    assert len(ck["g_optim"]["param_groups"][0]["params"]) == n_all              # reference indexing
    b = train.Trainer(**kw)
    checkpoint.load_checkpoint(path, b, map_location=dev)
    assert b.iteration == 3
    for (n, p), (_, q) in zip(a.generator.named_parameters(), b.generator.named_parameters()):
        assert torch.equal(p, q), n
    used = [n for n, _ in a.generator.named_parameters() if n not in a.frozen]
    sb = b.g_optim.state_dict()["state"]
    sa = a.g_optim.state_dict()["state"]
    assert len(sb) == len(sa) == len(used)
    for i in range(len(used)):
        assert torch.equal(sb[i]["exp_avg_sq"], sa[i]["exp_avg_sq"]) and float(sb[i]["step"]) == float(sa[i]["step"])
    b.step(data.batch(4))                                                        # the eager trainer continues
    path2 = checkpoint.save_checkpoint(str(tmp_path / checkpoint.checkpoint_name(4)), b)
    c = graph_train.GraphedTrainer(batch=4, **kw)
    checkpoint.load_checkpoint(path2, c, map_location=dev)
    assert c.iteration == 4 and float(c.g_optim.step_t) == float(b.g_optim.state_dict()["state"][0]["step"])
    for (n, p), (_, q) in zip(b.discriminator.named_parameters(), c.discriminator.named_parameters()):
        assert torch.equal(p, q), n
    log = c.step(data.batch(4))
    assert all(np.isfinite(v) for v in log.values())


def test_first_pass_of_the_regularisers_skips_unused_gradients_without_changing_any_value(monkeypatch):
    """op._dispatch.wanted: the recorded first pass of the path-length regulariser (gradient of the image w.r.t. latents
    and maps only, reference train.py:118-134) and of R1 (w.r.t. the real images, train.py:110-116) launches no weight /
    bias gradient kernel — every parameter gradient of the penalties is bit-identical to the unpruned passes, with
    fewer launches."""
    from torch.profiler import ProfilerActivity, profile

    from stylerenderer_amd.model import Discriminator, GeneratorWithMap

    dev = torch.device("cuda")
    torch.manual_seed(5)
    g = GeneratorWithMap(64, 64, 2).to(dev)
    d = Discriminator(64).to(dev)
    m = train.synthetic_mesh(2, dev, seed=3, face_sized=False)
    mesh = (m[0].requires_grad_(True), m[1].requires_grad_(True), m[2])      # like train.py:242-244
    z = torch.randn(2, 64, device=dev)
    probe = torch.randn(2, 3, 64, 64, device=dev)
    real = torch.randn(4, 3, 64, 64, device=dev)

    def run(prune):
        monkeypatch.setenv("SR_PRUNE_GRADS", "1" if prune else "0")
        for net in (g, d):
            for p in net.parameters():
                p.grad = None
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            img, latents, maps = g([z], mesh, return_latents=True, return_normals=True, randomize_noise=False)
            pen, _, lengths = train.g_path_regularize(img, [latents] + list(maps), 0.0, noise=probe)
            pen.backward()
            x = real.clone().requires_grad_(True)
            train.d_r1_loss(d(x), x).backward()
            torch.cuda.synchronize()
        launches = sum(e.count for e in prof.key_averages() if e.device_type is not None and "Memcpy" not in e.key)
        grads = {("g." if net is g else "d.") + n: p.grad.clone() for net in (g, d) for n, p in net.named_parameters()
                 if p.grad is not None}
        return grads, lengths.detach().clone(), launches

    full, len_full, n_full = run(False)
    lean, len_lean, n_lean = run(True)
    assert torch.equal(len_full, len_lean)
    assert set(full) == set(lean) and len(full) > 40
    differing = [k for k in full if not torch.equal(full[k], lean[k])]
    assert not differing, differing
    assert n_lean < n_full, (n_lean, n_full)
