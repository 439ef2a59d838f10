#!/usr/bin/env python3
"""GPU box: is the path-length phase's hipGraph replay reproducible, and does it equal the eager body?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
use_mesh = (sys.argv[2] != "nomesh") if len(sys.argv) > 2 else True
dev = torch.device("cuda")
faces = train.SyntheticFaceSource(dev, seed=0) if use_mesh else None
tr = graph_train.GraphedTrainer(size=size, latent=512, n_mlp=8, channel_multiplier=2, use_mesh=use_mesh, device=dev,
                                seed=0, batch=4, mesh_vertices=faces.model.dim[2] // 3 if use_mesh else None)
data = train.SyntheticImages(16, size, dev)
for _ in range(2):
    tr.step(data.batch(4), faces=faces)
mpl = tr.mean_path_length.clone()
state = torch.cuda.get_rng_state(dev)
flat = tr.flat_g
ref = None
for key in ("eager", "graph", "graph", "graph", "eager", "scramble", "graph", "eager", "graph"):
    if key == "scramble":
        tr._bodies()["g"]()
        tr.graphs["d"].replay()
        continue
    tr.mean_path_length.copy_(mpl)
    torch.cuda.set_rng_state(state, dev)
    flat.zero_()
    if key == "eager":
        tr._bodies()["path"]()
    else:
        tr.graphs["path"].replay()
    torch.cuda.synchronize()
    if ref is None:
        ref = flat.clone()
    print("%-6s path %.6f  path_length %.6f  mean_path %.6f  |flat| %.6e  max|flat - first eager| %.3e  rng offset after %d" % (
        key, float(tr.s_loss["path"]), float(tr.s_loss["path_length"]), float(tr.s_loss["mean_path"]),
        float(flat.double().norm()), float((flat - ref).abs().max()),
        int.from_bytes(bytes(torch.cuda.get_rng_state(dev)[8:16].tolist()), "little")))
