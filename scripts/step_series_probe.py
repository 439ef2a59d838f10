"""GPU box: per-step device time of the headline step (Generator(256) fwd+bwd, batch 16) over a long run — is the
10-step default of bench.py in a start-up transient?  usage: python scripts/step_series_probe.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SR_STRICT_NATIVE", "1")
import torch  # noqa: E402

from stylerenderer_amd import model  # noqa: E402
from stylerenderer_amd import distributed as sr_dist  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = model.Generator(256, 512, 8, channel_multiplier=2).to(dev)
sr_dist.freeze_unused_tail(g)
gen = torch.Generator(device=dev).manual_seed(1234)


def step():
    z = torch.randn(16, 512, device=dev, generator=gen)
    for p_ in g.parameters():
        p_.grad = None
    img, _ = g([z])
    img.sum().backward()


ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
t0 = time.perf_counter()
ev[0].record()
for i in range(n):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("wall %.1f ms/step over %d steps (incl. the cold first ones)" % (wall / n * 1e3, n))
for lo in range(0, n, 10):
    seg = ms[lo:lo + 10]
    print("steps %3d-%3d: mean %.3f  min %.3f  max %.3f" % (lo, lo + len(seg) - 1, sum(seg) / len(seg), min(seg), max(seg)))
