"""One-rank NCCL (RCCL) DistributedDataParallel run of the benchmark step: checks that the DDP wrapping used by
bench.py for N > 1 (bucket views, frozen ToRGB tail, grads set to None every step) works with the custom
autograd nodes.  usage (GPU box): python scripts/ddp_single_rank_check.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylerenderer_amd import model  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
g = model.Generator(256, 512, 8, channel_multiplier=2).to(dev)
for m in list(g.to_rgbs)[len(g.to_rgbs) // 2:]:
    for p in m.parameters():
        p.requires_grad_(False)
net = torch.nn.parallel.DistributedDataParallel(g, device_ids=[0], broadcast_buffers=False, bucket_cap_mb=32,
                                                gradient_as_bucket_view=True)


def step():
    z = torch.randn(16, 512, device=dev)
    for p in g.parameters():
        p.grad = None
    img, _ = net([z])
    img.sum().backward()


for _ in range(3):
    step()
dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
dist.barrier()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
missing = [n for n, p in g.named_parameters() if p.requires_grad and p.grad is None]
print("DDP(1 rank): %.2f ms/step = %.1f img/s; parameters without grad: %d" % (dt * 1e3, 16 / dt, len(missing)))
dist.destroy_process_group()
