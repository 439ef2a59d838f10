#!/usr/bin/env python3
"""GPU box: BASELINE config[3] rasterizer alone (rocprofv3 kernel traces / PMC passes of the raster kernels).
usage: python scripts/raster_probe.py [iters=20] [batch=64] [res=256]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import stylerenderer_amd.op as op  # noqa: E402
from stylerenderer_amd import synth  # noqa: E402

rz = importlib.import_module("stylerenderer_amd.op.rasterize")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
res = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
v0, tri = synth.face_sized_mesh()
vh = synth.random_poses(v0, batch, seed=1234)
v = torch.from_numpy(vh).to(dev)
nrm = torch.from_numpy(synth.vertex_normals(vh, tri)).to(dev)
t = torch.from_numpy(tri).to(dev)
vg, ng = v.clone().requires_grad_(), nrm.clone().requires_grad_()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fb():
    vg.grad = ng.grad = None
    op.rasterize(vg, ng, t, res).sum().backward()


print("api forward  %.4f ms" % timed(lambda: rz.forward(v, t, res, res, False, 1e-6), iters))
print("fused fwd    %.4f ms" % timed(lambda: op.rasterize(v, nrm, t, res), iters))
print("fwd + bwd    %.4f ms" % timed(fb, iters))
