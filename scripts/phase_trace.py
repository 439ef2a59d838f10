#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace: replays ONE captured phase of the config[2] iteration.
usage: python scripts/phase_trace.py <d|g|path|r1> [replays=8]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=4,
                                mesh_vertices=faces.model.dim[2] // 3)
data = train.SyntheticImages(64, 256, dev)
tr.step(data.batch(4), faces=faces, log=False)
torch.cuda.synchronize()
import time  # noqa: E402
time.sleep(1.0)                      # a gap in the trace: scripts/trace_summary.py --after-gap keeps what follows
for _ in range(n):
    tr.graphs[name].replay()
torch.cuda.synchronize()
