#!/usr/bin/env python3
"""GPU box: device time of each captured phase of the BASELINE config[2] iteration (graph replays timed with events).
usage: python scripts/phase_times.py [batch=4]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=batch,
                                mesh_vertices=faces.model.dim[2] // 3)
data = train.SyntheticImages(64, 256, dev)
for _ in range(2):
    tr.step(data.batch(batch), faces=faces, log=False)
torch.cuda.synchronize()
tot = 0.0
weights = {"d": 1.0, "d_opt": 1.0 + 1.0 / 16, "r1": 1.0 / 16, "g": 1.0, "g_opt": 1.0 + 1.0 / 4, "path": 1.0 / 4}
for name, g in tr.graphs.items():
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    w = weights.get(name, 1.0)
    tot += ms * w
    print("phase %-6s %7.2f ms per replay  x %.3f per iteration = %6.2f ms" % (name, ms, w, ms * w), flush=True)
print("sum over an average iteration: %.2f ms" % tot)
