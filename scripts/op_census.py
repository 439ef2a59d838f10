#!/usr/bin/env python3
"""GPU box: which operators make up one BASELINE config[2] iteration (eager), by count and device time.
usage: python scripts/op_census.py [phase]   phase in {all, d, g, path, r1}"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda", 0)
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=4,
                                mesh_vertices=faces.model.dim[2] // 3, capture=False)
data = train.SyntheticImages(16, 256, dev)
tr.step(data.batch(4), faces=faces, log=False)
tr._load_inputs(data.batch(4), None, faces)
bodies = tr._bodies()
names = {"all": ("d", "d_opt", "g", "g_opt"), "d": ("d",), "g": ("g",), "path": ("path",), "r1": ("r1",)}[phase]
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for n in names:
        bodies[n]()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.key.startswith("aten::") and e.device_time_total > 0]
rows = sorted(rows, key=lambda e: -e.device_time_total)
print("phase %s: aten operators with device time, by (op, input shapes)" % phase)
tot = sum(e.device_time_total for e in rows)
print("total aten device time %.1f us in %d calls" % (tot, sum(e.count for e in rows)))
for e in rows[:48]:
    print("%5d  %8.1f us  %-22s %s" % (e.count, e.device_time_total, e.key, str(e.input_shapes)[:110]))
by_name = {}
for e in rows:
    a = by_name.setdefault(e.key, [0, 0.0])
    a[0] += e.count
    a[1] += e.device_time_total
print("by operator:")
for k, (c, t) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%5d  %8.1f us  %s" % (c, t, k))
