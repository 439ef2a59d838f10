"""Where the split-bf16 path moves GeneratorWithMap(256): forward outputs of every StyledMapConv and the gradient arriving
at every norm_to_style head, SR_CONV_SPLIT_BF16=1 against the exact kernels (relative to each tensor's max)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from stylerenderer_amd import model, synth  # noqa: E402
from test_model_cpu import noise_list  # noqa: E402
from util import GWM_CASES  # noqa: E402

sdim, nmlp, batch, zkey, nkey, salt = GWM_CASES[256]
dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
caps = []
MODES = ("0", "w", "g", "c", "t", "1")
for mode in MODES:
    os.environ["SR_CONV_SPLIT_BF16"] = mode
    g = model.GeneratorWithMap(256, sdim, nmlp)
    synth.fill_state_dict(g.state_dict(), salt=salt)
    g = g.to(dev)
    cap = {}
    for name, m in g.named_modules():
        if name.startswith("convs.") and name.count(".") == 1:
            m.register_forward_hook(lambda mod, i, o, name=name: cap.__setitem__("fwd " + name, o.detach().clone()))
            m.register_full_backward_hook(lambda mod, gi, go, name=name: cap.__setitem__("gout " + name, go[0].detach().clone()))
        if name.startswith("norm_to_style.") and name.count(".") == 1:
            m.register_full_backward_hook(lambda mod, gi, go, name=name: cap.__setitem__("gout " + name, go[0].detach().clone()))
    v0, tri = synth.face_sized_mesh()
    v_np = synth.random_poses(v0, batch, seed=9)
    v, n = T(v_np).requires_grad_(), T(synth.vertex_normals(v_np, tri)).requires_grad_()
    img, lat, maps = g([T(synth.det_normal((batch, sdim), zkey))], (v, n, T(tri)), return_normals=True,
                       return_latents=True, noise=[x.to(dev) for x in noise_list(g, nkey)])
    proj = T(synth.det_normal(tuple(img.shape), zkey + 4))
    (img * proj).sum().backward()
    caps.append(cap)
for k in sorted(caps[0], key=lambda s: (s.split()[0], int(s.split(".")[1]))):
    a = caps[0][k].double()
    row = []
    for j in range(1, len(MODES)):
        b = caps[j][k].double()
        row.append("%s %.1e/%.1e" % (MODES[j], float((a - b).abs().max() / a.abs().max()),
                                     float((a - b).sum().abs() / a.abs().sum())))
    print("%-22s %s" % (k, "  ".join(row)))
print("(per family: max|d| / max|a|  /  |sum d| / sum|a|)")
