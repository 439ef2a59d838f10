#!/usr/bin/env python3
"""GPU box: fused rasterize forward at every resolution GeneratorWithMap uses, tiled vs global-key path (set SR_RASTER_TILED)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import stylerenderer_amd.op as op  # noqa: E402
from stylerenderer_amd import synth  # noqa: E402

dev = torch.device("cuda")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
v0, tri = synth.face_sized_mesh()
vh = synth.random_poses(v0, b, seed=1234)
v = torch.from_numpy(vh).to(dev)
nrm = torch.from_numpy(synth.vertex_normals(vh, tri)).to(dev)
t = torch.from_numpy(tri).to(dev)
out = []
for res in (4, 8, 16, 32, 64, 128, 256):
    for _ in range(3):
        op.rasterize(v, nrm, t, res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        op.rasterize(v, nrm, t, res)
    e1.record()
    torch.cuda.synchronize()
    out.append("%d: %.1f us" % (res, e0.elapsed_time(e1) / 20 * 1e3))
print("SR_RASTER_TILED=%s batch %d  " % (os.environ.get("SR_RASTER_TILED", "1"), b) + "  ".join(out))
