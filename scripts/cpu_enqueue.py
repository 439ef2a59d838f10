import sys, time, torch
sys.path.insert(0, '/root/repo')
from stylerenderer_amd import model
dev = torch.device('cuda', 0)
torch.manual_seed(0)
g = model.Generator(256, 512, 8, channel_multiplier=2).to(dev)
for m in list(g.to_rgbs)[len(g.to_rgbs)//2:]:
    for p in m.parameters(): p.requires_grad_(False)
def step():
    z = torch.randn(16, 512, device=dev)
    for p in g.parameters(): p.grad = None
    img, _ = g([z]); img.sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f ms/step, total %.1f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
