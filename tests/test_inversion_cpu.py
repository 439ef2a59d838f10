"""CPU: the latent-inversion loop (BASELINE config[4]) and the LPIPS-shaped metric at plumbing size."""
import numpy as np
import torch

from stylerenderer_amd import inversion, lpips, model, synth


def tiny_setup(device="cpu", with_map=True):
    torch.manual_seed(0)
    cls = model.GeneratorWithMap if with_map else model.Generator
    g = cls(16, 32, 2)
    synth.fill_state_dict(g.state_dict(), salt=91)
    g = g.to(device)
    v0, tri = synth.uv_ellipsoid(10, 12)
    v = torch.from_numpy(v0[None]).to(device)
    n = torch.from_numpy(synth.vertex_normals(v0[None], tri)).to(device)
    mesh = (v, n, torch.from_numpy(tri).to(device))
    return g, mesh


def test_pnetlin_structure_and_properties():
    net = lpips.PNetLin()
    assert net.chns == [64, 128, 256, 512, 512] and net.L == 5
    assert sum(len(s) for s in net.net.slices) == 13                       # VGG16: 13 convolutions
    a = torch.from_numpy(synth.det_uniform((2, 3, 32, 32), 1))
    b = torch.from_numpy(synth.det_uniform((2, 3, 32, 32), 2))
    d_ab, d_ba, d_aa = net(a, b), net(b, a), net(a, a)
    assert d_ab.shape == (2, 1, 1, 1) and torch.allclose(d_ab, d_ba, rtol=1e-5)
    assert float(d_aa.abs().max()) == 0.0 and float(d_ab.min()) > 0
    feats = net.features(a)
    assert [f.shape[1] for f in feats] == net.chns and [f.shape[2] for f in feats] == [32, 16, 8, 4, 2]
    assert torch.allclose((feats[0] ** 2).sum(1), torch.ones(2, 32, 32), atol=1e-4)     # unit-normalised channels
    # the real weights' key layout is accepted
    state = {"%d.%s" % (i, k): (torch.zeros_like(l.weight) if k == "weight" else torch.ones_like(l.bias))
             for ls, idx in zip(net.net.slices, lpips.VGG_FEATURE_INDEX) for l, i in zip(ls, idx) for k in ("weight", "bias")}
    net.net.load_trunk_state_dict(state)
    assert float(net.net.slices[0][0].weight.abs().max()) == 0.0
    net.load_lin_state_dict({"lin%d.model.1.weight" % k: torch.full((1, c, 1, 1), 0.5) for k, c in enumerate(net.chns)})
    assert float(net.lins[3].min()) == 0.5


def test_inversion_reduces_the_loss_and_moves_latent_and_pose():
    g, mesh = tiny_setup()
    net = lpips.PNetLin()
    with torch.no_grad():
        w_true = g.style(torch.from_numpy(synth.det_normal((1, 32), 5))).unsqueeze(1).repeat(1, g.n_latent, 1)
        noise = [torch.from_numpy(synth.det_normal((1, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)), 40 + i))
                 for i in range(g.num_layers)]
        rot = inversion.utils_3d.euler_mat(torch.tensor([[0.25, -0.1, 0.05]]), "yxz")[0]
        posed = (torch.matmul(mesh[0], rot) + torch.tensor([0.05, -0.03, 0.0]), torch.matmul(mesh[1], rot), mesh[2])
        target, _, _ = g([w_true], posed, input_is_latent=True, noise=noise)
    inv = inversion.LatentInverter(g, net, target, mesh, lr=0.05, pose_lr=0.02, noise=noise, n_mean_latent=64)
    hist = inv.run(25).numpy()
    assert np.isfinite(hist).all() and hist[-1] < 0.8 * hist[0]
    assert float(inv.pose.detach().abs().max()) > 1e-3                     # gradients reached the pose through the rasterizer
    assert float((inv.w.detach() - inv.w.detach()[:, :1]).abs().max()) > 0  # W+ rows evolve separately
    assert inv.image.shape == target.shape
