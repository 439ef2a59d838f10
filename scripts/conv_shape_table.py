#!/usr/bin/env python3
"""GPU box: per-shape table of the MFMA convolution / weight-gradient launches of the BASELINE config[2] iteration at
4 images per GPU — every phase run once eagerly under op.conv's launch hook (events on the launch stream).  Columns:
launches per phase, average ms, algorithmic TFLOP/s (direct-convolution flops), executed TFLOP/s (Winograd-eligible
stride-1 3x3 shapes issue 16/36 of the direct multiplies) and the executed fraction of the 157.3 TFLOP/s fp32 MFMA roof.
usage: python scripts/conv_shape_table.py [batch=4] > profiles/rNN_conv_shapes_b4.md"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from stylerenderer_amd import graph_train, train  # noqa: E402
from stylerenderer_amd.op import conv as conv_op  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=batch,
                                mesh_vertices=faces.model.dim[2] // 3, capture=False)
data = train.SyntheticImages(64, 256, dev)
tr.step(data.batch(batch), faces=faces, log=False)          # warm-up (lazy initialisation, caches)
torch.cuda.synchronize()
print("# MFMA convolution launches of the config[2] iteration, %d images per GPU (eager, per-launch events)\n" % batch)
for name in ("d", "r1", "g", "path"):
    conv_op.PROFILE = []
    tr._eager_phase(name)
    torch.cuda.synchronize()
    prof, conv_op.PROFILE = conv_op.PROFILE, None
    agg = collections.OrderedDict()
    for kind, geom, fl, e0, e1 in prof:
        a = agg.setdefault((kind,) + tuple(geom), [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += fl
        a[3] += bench.executed_flops(kind, geom, fl)
    tot_ms = sum(a[1] for a in agg.values())
    tot_exe = sum(a[3] for a in agg.values())
    print("## phase %s: %d launches, %.2f ms in MFMA kernels, %.1f TFLOP/s executed = %.2f of the roof\n"
          % (name, len(prof), tot_ms, tot_exe / tot_ms / 1e9, tot_exe / tot_ms / 1e9 / bench.FP32_MFMA_PEAK_TFLOPS))
    print("| kind | k | stride | transposed | B | Cin | Cout | grid | launches | avg ms | alg TF/s | exec TF/s | frac |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for key, (cnt, ms, fl, exe) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        kind, k, stride, tr_, b, c, n, gh, gw = key
        print("| %s | %d | %d | %d | %d | %d | %d | %dx%d | %d | %.3f | %.1f | %.1f | %.2f |" % (
            kind, k, stride, tr_, b, c, n, gh, gw, cnt, ms / cnt, fl / ms / 1e9, exe / ms / 1e9,
            exe / ms / 1e9 / bench.FP32_MFMA_PEAK_TFLOPS))
    print()
