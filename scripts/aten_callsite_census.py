#!/usr/bin/env python3
"""GPU box: which lines of this package issue the ATen launches of one config[2] phase (eager bodies)?
TorchDispatchMode + the innermost stylerenderer_amd frame of every dispatched op that launches a kernel; ops issued by
C++ autograd nodes (no Python frame) are grouped by name and shape.  usage: python scripts/aten_callsite_census.py [d|g|path|r1]"""
import collections
import os
import re
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "g"
dev = torch.device("cuda", 0)
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=4,
                                mesh_vertices=faces.model.dim[2] // 3, capture=False)
data = train.SyntheticImages(16, 256, dev)
tr.step(data.batch(4), faces=faces, log=False)
tr._load_inputs(data.batch(4), None, faces)
body = tr._bodies()[phase]

ANOMALY = os.environ.get("SR_CENSUS_ANOMALY", "0") == "1"
if ANOMALY:
    torch.autograd.set_detect_anomaly(True, check_nan=False)

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
            if site is None:
                shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
                node = torch._C._current_autograd_node()
                origin = ""
                if node is not None and ANOMALY:
                    # anomaly mode keeps the forward traceback of every node: the line of this package that CREATED the
                    # node whose backward issues this op (for the engine's fan-in adds: the node that produced the gradient)
                    tb = node.metadata.get("traceback_") or []
                    for line in reversed(tb):
                        m = re.search(r'File "([^"]*stylerenderer_amd[^"]*)", line (\d+), in (\w+)', line)
                        if m and "scripts" not in m.group(1):
                            origin = " <- %s:%s %s" % (os.path.relpath(m.group(1), ROOT), m.group(2), m.group(3))
                            break
                site = "(%s) %s%s" % (type(node).__name__ if node is not None else "C++ autograd node", shp, origin)
            sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


torch.cuda.synchronize()
with Census():
    body()
torch.cuda.synchronize()
print("phase %s: %d dispatched ATen ops that may launch" % (phase, sum(sites.values())))
for (name, site), n in sites.most_common(int(os.environ.get("SR_CENSUS_TOP", "70"))):
    print("%4d  %-28s %s" % (n, name, site))
