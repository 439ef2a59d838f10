"""GPU: latent inversion on the HIP path — hipGraph replay equals eager iterations, and two runs from the same
state are bit-identical (the rasterizer's gradient is a deterministic gather; round 1 scattered with float atomics)."""
import numpy as np
import pytest
import torch

from stylerenderer_amd import inversion, lpips, synth
from test_inversion_cpu import tiny_setup

pytestmark = pytest.mark.gpu


def make(use_graph):
    dev = torch.device("cuda")
    g, mesh = tiny_setup(dev)
    net = lpips.PNetLin().to(dev)
    noise = [torch.from_numpy(synth.det_normal((1, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)), 40 + i)).to(dev)
             for i in range(g.num_layers)]
    with torch.no_grad():
        w_true = g.style(torch.from_numpy(synth.det_normal((1, 32), 5)).to(dev)).unsqueeze(1).repeat(1, g.n_latent, 1)
        rot = inversion.utils_3d.euler_mat(torch.tensor([[0.25, -0.1, 0.05]], device=dev), "yxz")[0]
        posed = (torch.matmul(mesh[0], rot).contiguous(), torch.matmul(mesh[1], rot).contiguous(), mesh[2])
        target, _, _ = g([w_true], posed, input_is_latent=True, noise=noise)
    torch.manual_seed(3)                                      # mean_latent draws
    return inversion.LatentInverter(g, net, target, mesh, lr=0.05, pose_lr=0.02, noise=noise, n_mean_latent=64,
                                    use_graph=use_graph)


def test_perceptual_trunk_on_device_matches_cpu():
    net = lpips.PNetLin()
    a = torch.from_numpy(synth.det_uniform((2, 3, 64, 64), 1))
    b = torch.from_numpy(synth.det_uniform((2, 3, 64, 64), 2))
    want = net(a, b)
    ad = a.cuda().requires_grad_()
    got = net.cuda()(ad, b.cuda())
    assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    (ga,) = torch.autograd.grad(got.sum(), ad)
    ac = a.clone().requires_grad_()
    (gc,) = torch.autograd.grad(net.cpu()(ac, b).sum(), ac)
    # 13 ReLU layers: a handful of pre-activations change sign between two fp32 forward passes (see
    # test_model_gpu.test_full_gradient_tensors_gpu_vs_own_cpu_path), so the bar is on the gradient as a whole
    assert float((ga.cpu() - gc).norm() / gc.norm()) <= 2e-2


def test_graph_replay_equals_eager_and_is_reproducible():
    runs = {}
    for key, use_graph in (("eager", False), ("graph", True), ("graph2", True)):
        inv = make(use_graph)
        runs[key] = (inv.run(24).cpu().numpy(), inv.w.detach().cpu().clone(), inv.pose.detach().cpu().clone(),
                     inv.graph is not None)
    assert runs["graph"][3] and not runs["eager"][3]
    for h, _, _, _ in runs.values():
        assert np.isfinite(h).all() and h[-1] < 0.8 * h[0]
    # same kernels, same order: the captured iteration computes exactly what the eager one does
    assert np.allclose(runs["graph"][0], runs["eager"][0], rtol=1e-4, atol=1e-6)
    # run-to-run: bit-identical history, latent and pose
    assert np.array_equal(runs["graph"][0], runs["graph2"][0])
    assert torch.equal(runs["graph"][1], runs["graph2"][1]) and torch.equal(runs["graph"][2], runs["graph2"][2])
    assert float(runs["graph"][2].abs().max()) > 1e-3


def test_lpips_on_device_matches_the_reference_fixture(golden):
    """tests/golden/lpips_vgg.npz = the reference's PNetLin (networks_basic.py:27-112) with the real v0.1 heads."""
    gold = golden("lpips_vgg")
    net = lpips.PNetLin().cuda()
    in0 = torch.from_numpy(gold["in0"]).cuda().requires_grad_(True)
    val, res = net(in0, torch.from_numpy(gold["in1"]).cuda(), retPerLayer=True)
    want = gold["value"]
    e = float(np.abs(val.detach().cpu().numpy() - want).max() / np.abs(want).max())
    assert e < 5e-5, e
    per = np.stack([r.detach().cpu().numpy().reshape(-1) for r in res], 0)
    e = float(np.abs(per[1:] - gold["per_layer"][1:]).max() / np.abs(gold["per_layer"][1:]).max())
    assert e < 5e-5, e
    (g,) = torch.autograd.grad(val.sum(), in0)
    # 13 ReLU layers: a few pre-activations flip sign between two fp32 evaluation orders; bar on the whole gradient
    e = float(np.linalg.norm(g.cpu().numpy() - gold["grad_in0"]) / np.linalg.norm(gold["grad_in0"]))
    assert e < 2e-2, e


# ---------------------------------------------------------------------------------------------------
# BASELINE config[4] at its size: GeneratorWithMap(256, 512, 8) + the 24 962-vertex mesh + LPIPS(VGG16), as bench.py's
# inversion leg builds it.
def full_size_setup(use_graph):
    from stylerenderer_amd import model

    dev = torch.device("cuda")
    torch.manual_seed(0)
    g = model.GeneratorWithMap(256, 512, 8, channel_multiplier=2).to(dev)
    net = lpips.PNetLin().to(dev)
    v0, tri = synth.face_sized_mesh()
    v = torch.from_numpy(v0[None]).to(dev)
    nrm = torch.from_numpy(synth.vertex_normals(v0[None], tri)).to(dev)
    mesh = (v, nrm, torch.from_numpy(tri).to(dev))
    with torch.no_grad():
        w_true = g.style(torch.randn(1, 512, device=dev)).unsqueeze(1).repeat(1, g.n_latent, 1)
        rot = inversion.utils_3d.euler_mat(torch.tensor([[0.3, -0.1, 0.05]], device=dev), "yxz")[0]
        posed = (torch.matmul(v, rot).contiguous(), torch.matmul(nrm, rot).contiguous(), mesh[2])
        noise = [n.detach() for n in g.make_noise()]
        target, _, _ = g([w_true], posed, input_is_latent=True, noise=noise)
    torch.manual_seed(5)                                      # mean_latent draws
    return inversion.LatentInverter(g, net, target, mesh, noise=noise, use_graph=use_graph)


def test_config4_full_size_graph_equals_eager_is_reproducible_and_converges():
    steps = 12
    runs = {}
    for key, use_graph in (("eager", False), ("graph", True), ("graph2", True)):
        inv = full_size_setup(use_graph)
        hist = inv.run(steps)
        runs[key] = (hist.cpu().numpy(), inv.w.detach().cpu().clone(), inv.pose.detach().cpu().clone())
        if key == "graph":
            assert inv.graph is not None
            # the pose gradient exists only through sr_rasterize_grad (the vertices reach the image through the
            # rasterised normal maps alone)
            assert inv.pose.grad is not None and float(inv.pose.grad.abs().max()) > 0
            assert torch.isfinite(inv.pose.grad).all() and torch.isfinite(inv.w.grad).all()
            more = inv.run(50).cpu().numpy()                  # steps 13..62 of the same trajectory
            assert np.isfinite(more).all()
            assert more[-1] < 0.8 * runs[key][0][0], (runs[key][0][0], more[-1])
        del inv
        torch.cuda.empty_cache()
    for h, _, _ in runs.values():
        assert np.isfinite(h).all()
    err = np.abs(runs["graph"][0] - runs["eager"][0]) / np.abs(runs["eager"][0])
    assert err.max() <= 1e-4, err
    assert np.array_equal(runs["graph"][0], runs["graph2"][0])
    assert torch.equal(runs["graph"][1], runs["graph2"][1]) and torch.equal(runs["graph"][2], runs["graph2"][2])
    assert float(runs["graph"][2].abs().max()) > 1e-4        # the pose moved


def test_fused_lpips_layer_matches_the_operator_chain():
    """op.lpips_layer (normalisation, squared difference, lin, spatial mean in one launch each way) against the chain of
    tensor operators the reference spells out (networks_basic.py:62-85): value and gradient, per-sample and shared target."""
    from stylerenderer_amd import lpips
    from stylerenderer_amd.op.lpips_layer import lpips_layer

    DEV = "cuda"
    g = torch.Generator().manual_seed(8)
    for b, c, h, w, tb in ((1, 64, 32, 32, 1), (3, 128, 9, 7, 3), (2, 512, 16, 16, 1)):
        f0 = torch.relu(torch.randn(b, c, h, w, generator=g)).to(DEV).requires_grad_()
        t = lpips.normalize_tensor(torch.relu(torch.randn(tb, c, h, w, generator=g)).to(DEV))
        lin = torch.rand(1, c, 1, 1, generator=g).to(DEV)
        gd = torch.randn(b, 1, 1, 1, generator=g).to(DEV)
        want = lpips.spatial_average((((lpips.normalize_tensor(f0) - t) ** 2) * lin).sum(1, keepdim=True))
        (gw,) = torch.autograd.grad(want, f0, gd)
        got = lpips_layer(f0, t, lin)
        (gg,) = torch.autograd.grad(got, f0, gd)
        assert got.shape == want.shape
        assert torch.allclose(got, want, rtol=2e-5, atol=1e-9), (b, c, float((got - want).abs().max()))
        assert float((gg - gw).abs().max()) <= 2e-5 * float(gw.abs().max()) + 1e-12, (b, c)
        again = lpips_layer(f0, t, lin)
        assert torch.equal(got, again)                       # fixed-order reduction: run-to-run identical
