"""One-rank NCCL (RCCL) run of the hipGraph-replayed training iteration with the gradient all-reduce forced on
(force_collectives=True on a 1-rank group): checks that the bucketed RCCL collectives released by the event-record
nodes inside the replayed graphs order correctly and that the step still trains; prints when each bucket finished
relative to the end of its phase's replay.  Also runs the eager DDP Trainer on the same group.
usage (GPU box): python scripts/graph_ddp_single_rank_check.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylerenderer_amd import graph_train, train  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
# 0) what bench.py does first for N > 1: a DDP-wrapped generator leg (bucketed all-reduces in flight), destroyed
#    just before the graphs of the training leg are captured — the RCCL watchdog is busy at that point
from stylerenderer_amd import model  # noqa: E402

g = model.Generator(256, 512, 8, channel_multiplier=2).to(dev)
sr_dist_mod = __import__("stylerenderer_amd.distributed", fromlist=["x"])
sr_dist_mod.freeze_unused_tail(g)
net = torch.nn.parallel.DistributedDataParallel(g, device_ids=[0], broadcast_buffers=False, bucket_cap_mb=32,
                                                gradient_as_bucket_view=True)
for _ in range(3):
    for p in g.parameters():
        p.grad = None
    img, _ = net([torch.randn(4, 512, device=dev)])
    img.sum().backward()
del net, g, img
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=4,
                                mesh_vertices=faces.model.dim[2] // 3, force_collectives=True)
from stylerenderer_amd import distributed as sr_dist  # noqa: E402

print("reducer:", tr.reduce_g.describe(), tr.reduce_d.describe())
# the in-place reduce-scatter + all-gather form on the same (one-rank) group: must leave the buffer unchanged
probe = sr_dist.FlatGradReducer(tr.flat_g, world=1)
probe.world, probe.mode = 1, "rsag"
before = tr.flat_g.clone().normal_()
tr.flat_g.copy_(before)
probe()
torch.cuda.synchronize()
assert torch.equal(tr.flat_g, before), "in-place reduce-scatter + all-gather changed a one-rank buffer"
print("in-place reduce_scatter_tensor + all_gather_into_tensor: ok")
data = train.SyntheticImages(16, 256, dev)
for _ in range(2):
    tr.step(data.batch(4), faces=faces, log=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    out = tr.step(data.batch(4), faces=faces, log=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 8
vals = {k: float(v) for k, v in out.items()}
assert all(v == v for v in vals.values()), vals
print("graph replay + bucketed RCCL all-reduce released by in-graph events: %.1f ms/iter, losses %s" % (dt * 1e3, vals))
for name in ("d", "r1", "g", "path"):
    print(tr.measure_overlap(name))
dist.barrier()
dist.destroy_process_group()
