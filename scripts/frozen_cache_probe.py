#!/usr/bin/env python3
"""GPU box: generator forward + latent gradient with and without the frozen-weight preparation cache (must be identical)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import model, synth  # noqa: E402
from stylerenderer_amd.op.weight_prep import freeze_prepared_weights  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
g = model.GeneratorWithMap(64, 512, 8).to(dev).eval()
for p in g.parameters():
    p.requires_grad_(False)
v0, tri = synth.uv_ellipsoid(16, 14)
v = torch.from_numpy(v0[None]).to(dev)
n = torch.from_numpy(synth.vertex_normals(v0[None], tri)).to(dev)
mesh = (v, n, torch.from_numpy(tri).to(dev))
noise = [x.detach() for x in g.make_noise()]
w = torch.randn(1, g.n_latent, 512, device=dev)


def run():
    out = []
    for _ in range(3):
        ww = w.clone().requires_grad_(True)
        img, _, _ = g([ww], mesh, input_is_latent=True, noise=noise)
        (gw,) = torch.autograd.grad((img * img).sum(), ww)
        out.append((img.detach().clone(), gw.clone()))
    return out


a = run()
freeze_prepared_weights(g)
b = run()
for k, ((ia, ga), (ib, gb)) in enumerate(zip(a, b)):
    print("pass %d: image max diff %.3e (scale %.3e); latent-gradient max diff %.3e (scale %.3e)" % (
        k, float((ia - ib).abs().max()), float(ia.abs().max()), float((ga - gb).abs().max()), float(ga.abs().max())))
