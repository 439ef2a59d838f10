"""Counts leaky-ReLU sign-pattern differences between the CPU path and the HIP path of GeneratorWithMap(64) for a
range of latent keys (used to choose a fixture latent on which the two fp32 forward passes take the same side of
every kink; see tests/test_model_gpu.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from stylerenderer_amd import model, synth  # noqa: E402
from test_model_cpu import noise_list  # noqa: E402

size, sdim, batch, nkey, salt = 64, 64, 1, 5700, 53
v0, tri = synth.uv_ellipsoid(28, 24)
vh = synth.random_poses(v0, batch, seed=7)
nh = synth.vertex_normals(vh, tri)
for zkey in range(int(sys.argv[1]), int(sys.argv[2])):
    signs = {}
    for dev in ("cpu", "cuda"):
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        g = model.GeneratorWithMap(size, sdim, 2)
        synth.fill_state_dict(g.state_dict(), salt=salt)
        g = g.to(dev)
        st = {}
        for name, m in g.named_modules():
            if isinstance(m, (model.StyledConv, model.StyledMapConv)):
                m.register_forward_hook(lambda mod, i, o, name=name: st.__setitem__(name, (o.detach() > 0).cpu()))
        with torch.no_grad():
            g([T(synth.det_normal((batch, sdim), zkey))], (T(vh), T(nh), T(tri)), noise=[x.to(dev) for x in noise_list(g, nkey)])
        signs[dev] = st
    print("zkey", zkey, "flips", {k: int((signs["cpu"][k] != signs["cuda"][k]).sum()) for k in signs["cpu"]
                                  if int((signs["cpu"][k] != signs["cuda"][k]).sum())})
