#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace: BASELINE config[4] — capture the inversion step, sleep (a gap in the trace:
scripts/trace_summary.py --after-gap keeps what follows), then replay it N times.  usage: ... [replays=40]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import inversion, lpips, model, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = model.GeneratorWithMap(256, 512, 8, channel_multiplier=2).to(dev)
net = lpips.PNetLin().to(dev)
v0, tri = synth.face_sized_mesh()
v = torch.from_numpy(v0[None]).to(dev)
nrm = torch.from_numpy(synth.vertex_normals(v0[None], tri)).to(dev)
mesh = (v, nrm, torch.from_numpy(tri).to(dev))
with torch.no_grad():
    w_true = g.style(torch.randn(1, 512, device=dev)).unsqueeze(1).repeat(1, g.n_latent, 1)
    noise = [x.detach() for x in g.make_noise()]
    target, _, _ = g([w_true], mesh, input_is_latent=True, noise=noise)
inv = inversion.LatentInverter(g, net, target, mesh, noise=noise, use_graph=True)
inv.run(8)                            # warm-up + capture + a few replays
torch.cuda.synchronize()
print("kernel nodes per step:", inv.graph.kernel_nodes)
time.sleep(1.0)
t0 = time.perf_counter()
for _ in range(n):
    inv.graph.replay()
torch.cuda.synchronize()
print("replay %.3f ms/step" % ((time.perf_counter() - t0) / n * 1e3))
