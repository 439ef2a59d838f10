"""Per-tensor error of GeneratorWithMap(256)'s first-order parameter gradients against the float64 truth of
tests/golden/generator_map_s256.npz, exact-fp32 kernels vs SR_CONV_SPLIT_BF16=1 (which tensors the split path moves)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from stylerenderer_amd import model, synth  # noqa: E402
from test_model_cpu import noise_list  # noqa: E402
from util import GWM_CASES, KinkForcer  # noqa: E402

gold = np.load(os.path.join(ROOT, "tests", "golden", "generator_map_s256.npz"))
sdim, nmlp, batch, zkey, nkey, salt = GWM_CASES[256]
dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
res = {}
for mode in ("0", "1"):
    os.environ["SR_CONV_SPLIT_BF16"] = mode
    g = model.GeneratorWithMap(256, sdim, nmlp)
    synth.fill_state_dict(g.state_dict(), salt=salt)
    g = g.to(dev)
    v0, tri = synth.face_sized_mesh()
    v_np = synth.random_poses(v0, batch, seed=9)
    v, n = T(v_np).requires_grad_(), T(synth.vertex_normals(v_np, tri)).requires_grad_()
    forcer = KinkForcer(g, gold)
    img, lat, maps = g([T(synth.det_normal((batch, sdim), zkey))], (v, n, T(tri)), return_normals=True,
                       return_latents=True, noise=[x.to(dev) for x in noise_list(g, nkey)])
    forcer.close()
    proj = T(synth.det_normal(tuple(img.shape), zkey + 4))
    params = dict(g.named_parameters())
    grads = torch.autograd.grad((img * proj).sum(), list(params.values()), allow_unused=True)
    got = {k: x for k, x in zip(params, grads) if x is not None}
    names, t, o = gold["grad_names"], gold["grad_samples_f64"], gold["grad_sample_offsets"]
    for i, nme in enumerate(names):
        gg = got[str(nme)].reshape(-1).double().cpu().numpy()
        s = gg[synth.sample_index(gg.size, 256)]
        w = t[o[i]:o[i + 1]]
        res.setdefault(str(nme), []).append(float(np.abs(s - w).max() / max(np.abs(w).max(), 1e-30)))
rows = sorted(res.items(), key=lambda kv: -kv[1][1])
for k, (a, b) in rows[:25]:
    print("%-44s exact %.2e  split %.2e  x%.1f" % (k, a, b, b / max(a, 1e-30)))
