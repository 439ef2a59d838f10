"""cProfile of the host side of the benchmark step (enqueue only)."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, '/root/repo')
from stylerenderer_amd import model  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
g = model.Generator(256, 512, 8, channel_multiplier=2).to(dev)
for m in list(g.to_rgbs)[len(g.to_rgbs) // 2:]:
    for p in m.parameters():
        p.requires_grad_(False)


def step():
    z = torch.randn(16, 512, device=dev)
    for p in g.parameters():
        p.grad = None
    img, _ = g([z])
    img.sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
