"""Per-tensor error of Generator(256) gradients (HIP path, and the CPU path on this host) against
tests/golden/generator_s256.npz, plus the LeakyReLU sign-pattern differences between the two paths."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from stylerenderer_amd import model, synth  # noqa: E402
from test_model_cpu import noise_list  # noqa: E402

gold = np.load(os.path.join(ROOT, "tests", "golden", "generator_s256.npz"))
zkey = int(sys.argv[1]) if len(sys.argv) > 1 else 42
devs = sys.argv[2:] or ["cuda"]
signs = {}
for dev in devs:
    g = model.Generator(256, 512, 8)
    synth.fill_state_dict(g.state_dict(), salt=41)
    g = g.to(dev)
    st = {}
    for name, m in g.named_modules():
        if isinstance(m, model.StyledConv):
            m.register_forward_hook(lambda mod, i, o, name=name: st.__setitem__(name, (o.detach() > 0).cpu()))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    img, lat = g([T(synth.det_normal((1, 512), zkey))], return_latents=True, noise=[n.to(dev) for n in noise_list(g, 4300)])
    signs[dev] = st
    if zkey != 42:
        continue
    print(dev, "img", float(np.abs(img.detach().cpu().numpy() - gold["image"]).max() / np.abs(gold["image"]).max()))
    proj = T(synth.det_normal(tuple(img.shape), 46))
    params = dict(g.named_parameters())
    grads = torch.autograd.grad((img * proj).sum(), list(params.values()) + [lat], allow_unused=True)
    got = {n: x for n, x in zip(params, grads[:-1]) if x is not None}
    offs = gold["grad_sample_offsets"]
    for i, n in enumerate(gold["grad_names"]):
        want = gold["grad_samples"][offs[i]:offs[i + 1]].astype(np.float64)
        a = got[n].detach().reshape(-1).cpu().numpy().astype(np.float64)
        a = a[synth.sample_index(a.size, 256)]
        err = np.abs(a - want).max() / max(np.abs(want).max(), 1e-12)
        if err > 5e-6:
            print("  %-40s err %.3e scale %.3e numel %d" % (n, err, np.abs(want).max(), got[n].numel()))
    print(dev, "latent grad", float(np.abs(grads[-1].cpu().numpy() - gold["grad_latent"]).max() / np.abs(gold["grad_latent"]).max()))
if len(devs) == 2:
    a, b = signs[devs[0]], signs[devs[1]]
    print("zkey", zkey, "flips", {k: int((a[k] != b[k]).sum()) for k in a if int((a[k] != b[k]).sum())})
