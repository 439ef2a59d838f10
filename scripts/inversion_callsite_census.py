#!/usr/bin/env python3
"""GPU box: which lines of this package issue the ATen launches of ONE eager BASELINE config[4] inversion step
(TorchDispatchMode + innermost stylerenderer_amd frame; ops of C++ autograd nodes by name and shape)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from stylerenderer_amd import inversion, lpips, model, synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = model.GeneratorWithMap(256, 512, 8, channel_multiplier=2).to(dev)
net = lpips.PNetLin().to(dev)
v0, tri = synth.face_sized_mesh()
v = torch.from_numpy(v0[None]).to(dev)
nrm = torch.from_numpy(synth.vertex_normals(v0[None], tri)).to(dev)
mesh = (v, nrm, torch.from_numpy(tri).to(dev))
with torch.no_grad():
    noise = [n.detach() for n in g.make_noise()]
    w_true = g.style(torch.randn(1, 512, device=dev)).unsqueeze(1).repeat(1, g.n_latent, 1)
    target, _, _ = g([w_true], mesh, input_is_latent=True, noise=noise)
inv = inversion.LatentInverter(g, net, target, mesh, noise=noise, use_graph=False)
inv.run(3)
torch.cuda.synchronize()

SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::expand", "aten::permute", "aten::transpose", "aten::t",
        "aten::select", "aten::slice", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::as_strided",
        "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::new_empty", "aten::split", "aten::unbind",
        "aten::split_with_sizes", "aten::view_as", "aten::is_", "aten::sym_", "aten::_local_scalar", "aten::lift",
        "aten::unsafe_split", "aten::chunk", "aten::narrow", "aten::stride", "aten::size", "aten::numel", "aten::set_",
        "aten::result_type", "aten::item", "aten::record_stream", "aten::is_pinned", "aten::contiguous")
sites = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if not name.startswith(SKIP):
            site = None
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "stylerenderer_amd" in fr.filename and "scripts" not in fr.filename:
                    site = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                    break
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            if site is None:
                site = "(C++ autograd node)"
            sites[(name, site, str(shp))] += 1
        return func(*args, **(kwargs or {}))


with Census():
    inv.run(1)
torch.cuda.synchronize()
print("inversion step: %d dispatched ATen ops that may launch" % sum(sites.values()))
for (name, site, shp), n in sites.most_common(90):
    print("%4d  %-26s %-64s %s" % (n, name, site, shp))
