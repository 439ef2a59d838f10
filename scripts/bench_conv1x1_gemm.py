#!/usr/bin/env python3
"""GPU box: the discriminator's 1x1 skip convolutions (and their data gradients: the same operator with the channels
swapped), GEMM-shaped kernel (csrc/conv1x1_gemm.hip) against the 1x1 instantiation of k_conv_mfma.
usage: python scripts/bench_conv1x1_gemm.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd.op.conv import conv2d_mfma  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (b, c, n, res) in ((8, 128, 256, 128), (8, 256, 128, 128), (8, 256, 512, 64), (8, 512, 256, 64), (8, 512, 512, 32),
                       (4, 128, 256, 128), (4, 256, 512, 64), (4, 512, 512, 32), (8, 512, 512, 16)):
    x = torch.randn(b, c, res, res, device=dev)
    wt = torch.randn(1, c, n, device=dev) / c ** 0.5
    out = {}
    for flag in ("0", "1", "force"):
        os.environ["SR_CONV1X1_GEMM"] = flag
        out[flag] = (timed(lambda: conv2d_mfma(x, wt, None, None, None, 1, 1, 0, False)), conv2d_mfma(x, wt, None, None, None, 1, 1, 0, False))
    gf = 2.0 * b * res * res * c * n / 1e9
    err = float((out["0"][1] - out["force"][1]).abs().max())
    print("B %d  %3d -> %3d  %3d^2  %6.2f GFLOP   window kernel %7.1f us (%5.1f TF)   default %7.1f us   GEMM forced %7.1f us (%5.1f TF)   max diff %.1e"
          % (b, c, n, res, gf, out["0"][0], gf / out["0"][0] * 1e3, out["1"][0], out["force"][0], gf / out["force"][0] * 1e3, err), flush=True)
