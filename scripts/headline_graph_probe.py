"""GPU box: the headline step (Generator(256) fwd+bwd, batch 16) launched eagerly vs replayed from one hipGraph, same
process, alternating blocks.  usage: python scripts/headline_graph_probe.py [steps per block]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SR_STRICT_NATIVE", "1")
import torch  # noqa: E402

from stylerenderer_amd import graphs, model  # noqa: E402
from stylerenderer_amd import distributed as sr_dist  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = model.Generator(256, 512, 8, channel_multiplier=2).to(dev)
sr_dist.freeze_unused_tail(g)


def step():
    z = torch.randn(16, 512, device=dev)
    for p_ in g.parameters():
        p_.grad = None
    img, _ = g([z])
    img.sum().backward()


def timed(fn, k):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3


for _ in range(4):
    step()
torch.cuda.synchronize()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(2):
        step()
torch.cuda.synchronize()
graph = graphs.capture(step)
print("graph: %d kernel nodes, %d memset nodes replaced" % (graph.kernel_nodes, graph.memset_nodes_replaced))
for r in range(3):
    e = timed(step, n)
    q = timed(graph.replay, n)
    print("block %d: eager %.3f ms/step (%.1f img/s)   graph replay %.3f ms/step (%.1f img/s)" % (r, e, 16e3 / e, q, 16e3 / q))
# host cost of enqueueing one eager step on an empty queue
torch.cuda.synchronize()
t = time.perf_counter()
step()
h = (time.perf_counter() - t) * 1e3
torch.cuda.synchronize()
print("host enqueue of one eager step: %.2f ms" % h)
