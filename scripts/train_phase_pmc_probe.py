#!/usr/bin/env python3
"""GPU box, under rocprofv3 --pmc: builds the config[2] trainer (one eager warm-up pass: SR_GRAPH_WARMUP=1), runs
iteration 0 (every phase once) and then replays every captured phase ONCE more in the order printed below — the last
sum(kernel nodes) dispatches of the process, which bench.graph_phase_traffic() slices per phase.
usage: python scripts/train_phase_pmc_probe.py [batch=4]"""
import os
import sys

os.environ.setdefault("SR_GRAPH_WARMUP", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

ORDER = ("d", "r1", "g", "path", "d_opt", "g_opt", "ema")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=batch,
                                mesh_vertices=faces.model.dim[2] // 3)
data = train.SyntheticImages(16, 256, dev)
tr.step(data.batch(batch), faces=faces, log=False)
torch.cuda.synchronize()
for name in ORDER:
    tr.graphs[name].replay()
    torch.cuda.synchronize()
print("phase order", ORDER, "kernel nodes", [tr.graphs[n].kernel_nodes for n in ORDER])
