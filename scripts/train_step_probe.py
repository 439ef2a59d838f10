#!/usr/bin/env python3
"""GPU box: the BASELINE config[2] iteration alone (for rocprofv3 traces and enqueue-vs-GPU accounting).
usage: python scripts/train_step_probe.py [iters=16] [batch=4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
faces = train.SyntheticFaceSource(dev, seed=0)
if os.environ.get("SR_TRAIN_GRAPHS", "1") != "0":
    tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=batch,
                                    mesh_vertices=faces.model.dim[2] // 3)
else:
    tr = train.Trainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0)
data = train.SyntheticImages(64, 256, dev)
for _ in range(2):
    tr.step(data.batch(batch), faces=faces, log=False)
tr.iteration = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    tr.step(data.batch(batch), faces=faces, log=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("train probe: %d iters batch %d: enqueue %.1f ms/iter, total %.1f ms/iter, %.1f img/s"
      % (iters, batch, (t1 - t0) / iters * 1e3, (t2 - t0) / iters * 1e3, batch * iters / (t2 - t0)))
