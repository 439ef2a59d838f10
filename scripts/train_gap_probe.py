#!/usr/bin/env python3
"""GPU box: where the G+D iteration spends time OUTSIDE its graph replays.  Events around every `_run`, the input
refresh, the EMA and the synthetic data source; prints per iteration the replay time, the time between replays and the
host's own time per step().   usage: python scripts/train_gap_probe.py [iters=16] [batch=4]"""
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
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=batch,
                                mesh_vertices=faces.model.dim[2] // 3)
data = train.SyntheticImages(64, 256, dev)
for _ in range(3):
    tr.step(data.batch(batch), faces=faces, log=False)
tr.iteration = 0
torch.cuda.synchronize()

marks = []          # (label, start event, end event)


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **k):
        e0 = ev()
        r = fn(*a, **k)
        marks.append((label(*a) if callable(label) else label, e0, ev()))
        return r
    setattr(obj, name, inner)


wrap(tr, "_run", lambda n: "run:" + n)
wrap(tr, "_load_inputs", "load_inputs")
host = []
first = ev()
t0 = time.perf_counter()
for _ in range(iters):
    e0 = ev()
    img = data.batch(batch)
    marks.append(("data.batch", e0, ev()))
    t1 = time.perf_counter()
    tr.step(img, faces=faces, log=False)
    host.append(time.perf_counter() - t1)
last = ev()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
total = first.elapsed_time(last)
acc_ms = {}
for lab, a, b in marks:
    acc_ms[lab] = acc_ms.get(lab, 0.0) + a.elapsed_time(b)
inside = sum(acc_ms.values())
print("iteration %.3f ms   (host enqueue %.2f ms/iter, step() alone %.2f)" % (total / iters, 1e3 * t_enq / iters,
                                                                             1e3 * sum(host) / iters))
for lab, v in sorted(acc_ms.items(), key=lambda t: -t[1]):
    print("  %-14s %8.3f ms/iter" % (lab, v / iters))
print("  %-14s %8.3f ms/iter" % ("(between)", (total - inside) / iters))
# gaps between consecutive marks
gaps = {}
for (l0, _, b0), (l1, a1, _) in zip(marks[:-1], marks[1:]):
    g = b0.elapsed_time(a1)
    key = l0 + " -> " + l1
    gaps[key] = gaps.get(key, 0.0) + g
for k, v in sorted(gaps.items(), key=lambda t: -t[1])[:10]:
    print("  gap %-34s %7.3f ms/iter" % (k, v / iters))
