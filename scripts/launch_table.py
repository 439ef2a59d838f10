#!/usr/bin/env python3
"""GPU box: every MFMA launch of one BASELINE config[1] step (Generator(256) forward + backward, batch 16) by
(kind, geometry): launches per step, ms per launch, algorithmic TFLOP/s.  usage: python scripts/launch_table.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import model  # noqa: E402
from stylerenderer_amd.op import conv as conv_op  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = model.Generator(256, 512, 8).to(dev)
z = torch.randn(16, 512, device=dev)


def step():
    img, _ = g([z])
    img.square().mean().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
steps = 4
conv_op.PROFILE = []
for _ in range(steps):
    step()
torch.cuda.synchronize()
prof, conv_op.PROFILE = conv_op.PROFILE, None
agg = collections.OrderedDict()
for kind, geom, fl, e0, e1 in prof:
    a = agg.setdefault((kind,) + tuple(geom), [0, 0.0, fl])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
print("%-6s %-44s %6s %9s %8s %9s" % ("kind", "(k, stride, transposed, B, C, N, GH, GW)", "n/step", "ms/launch", "TFLOP/s", "ms/step"))
tot = 0.0
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms / steps
    print("%-6s %-44s %6.1f %9.4f %8.1f %9.3f" % (key[0], str(key[1:]), n / steps, ms / n, fl / (ms / n * 1e-3) / 1e12, ms / steps))
print("total %.3f ms/step in MFMA launches (incl. their reductions / weight transforms)" % tot)
