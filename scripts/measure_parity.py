#!/usr/bin/env python3
"""GPU box: measured errors behind the tolerances written in tests/ (printed, not asserted)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_train_parity_cpu as tp  # noqa: E402
from stylerenderer_amd import layers, model, synth  # noqa: E402
from test_model_cpu import noise_list  # noqa: E402
from util import check_grad_samples, rel_err  # noqa: E402

T = torch.from_numpy
G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))  # noqa: E731
dev = "cuda"
print("train_step_s8 (r1 samples, pl samples):", tp.run_all(G("train_step_s8"), dev, 1, 1, 1))
gold = G("generator_s8")
g8 = model.Generator(8, 64, 2)
synth.fill_state_dict(g8.state_dict(), salt=41)
g8 = g8.to(dev)
z = T(synth.det_normal((2, 64), 42)).to(dev)
noise = [n.to(dev) for n in noise_list(g8, 4300)]
img, lat = g8([z], return_latents=True, noise=noise)
proj = T(synth.det_normal(tuple(img.shape), 46)).to(dev)
params = dict(g8.named_parameters())
grads = torch.autograd.grad((img * proj).sum(), list(params.values()), allow_unused=True)
got = {n: g for n, g in zip(params, grads) if g is not None}
print("generator_s8 grad samples:", check_grad_samples(got, gold["grad_names"], gold["grad_samples"],
                                                       gold["grad_sample_offsets"], 1))
img, lat = g8([z], return_latents=True, noise=noise)
pl_noise = T(synth.det_normal(tuple(img.shape), 47)).to(dev) / np.sqrt(img.shape[2] * img.shape[3])
(gl,) = torch.autograd.grad((img * pl_noise).sum(), lat, create_graph=True)
flat = gl.reshape(gl.shape[0], -1)
lengths = torch.sqrt((flat * flat).sum(1))
penalty = (lengths - 0.01 * lengths.mean()).pow(2).mean()
g8.zero_grad()
penalty.backward()
got = {n: p.grad for n, p in g8.named_parameters() if p.grad is not None}
print("generator_s8 pl grad samples:", check_grad_samples(got, gold["pl_grad_names"], gold["pl_grad_samples"],
                                                          gold["pl_grad_sample_offsets"], 1))
gold = G("discriminator_s16")
d = model.Discriminator(16)
synth.fill_state_dict(d.state_dict(), salt=61)
d = d.to(dev)
x = T(gold["x"]).to(dev).requires_grad_()
y = d(x)
(gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
r1 = (gx * gx).reshape(4, -1).sum(1).mean()
d.zero_grad()
r1.backward()
got = {n: p.grad for n, p in d.named_parameters() if p.grad is not None}
print("discriminator_s16 r1 grad samples:", check_grad_samples(got, gold["r1_grad_names"], gold["r1_grad_samples"],
                                                               gold["r1_grad_sample_offsets"], 1))
gold = G("generator_s256")
g = model.Generator(256, 512, 8)
synth.fill_state_dict(g.state_dict(), salt=41)
g = g.to(dev)
with torch.no_grad():
    img, _ = g([T(synth.det_normal((1, 512), 42)).to(dev)], noise=[n.to(dev) for n in noise_list(g, 4300)])
print("generator_s256 image rel err:", rel_err(img.cpu().numpy(), gold["image"]))
gold = G("modconv")
for tag, kw in (("plain", dict(in_channel=8, out_channel=6, kernel_size=3, style_dim=16)),
                ("up", dict(in_channel=8, out_channel=6, kernel_size=3, style_dim=16, upsample=True)),
                ("rgb", dict(in_channel=8, out_channel=3, kernel_size=1, style_dim=16, demodulate=False))):
    m = layers.ModulatedConv2d(**kw)
    synth.fill_state_dict(m.state_dict(), salt=31)
    m = m.to(dev)
    x = T(gold[tag + "_x"]).to(dev).requires_grad_()
    s = T(gold[tag + "_s"]).to(dev).requires_grad_()
    y = m(x, s)
    grads = torch.autograd.grad(y, [x, s, m.weight, m.modulation.weight, m.modulation.bias], T(gold[tag + "_gy"]).to(dev))
    errs = [rel_err(y.detach().cpu().numpy(), gold[tag + "_y"])] + [
        rel_err(a.cpu().numpy(), gold[tag + "_" + k]) for a, k in zip(grads, ("gx", "gs", "gw", "gmw", "gmb"))]
    print("modconv", tag, ["%.1e" % e for e in errs])
