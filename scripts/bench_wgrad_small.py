#!/usr/bin/env python3
"""GPU box: weight gradient of the map heads' 3x3 layers (3 -> 3, 3 -> 4 channels), streaming kernel against the MFMA tiles.
usage: python scripts/bench_wgrad_small.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd.op.conv import conv2d_wgrad_mfma  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=50):
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


for (b, c, n, res) in ((4, 3, 3, 256), (4, 3, 4, 256), (4, 3, 4, 128), (4, 3, 4, 64), (4, 3, 4, 32), (2, 3, 4, 256), (1, 3, 4, 256)):
    x = torch.randn(b, c, res, res, device=dev)
    gy = torch.randn(b, n, res, res, device=dev)
    out = {}
    for flag in ("0", "1"):
        os.environ["SR_WGRAD_SMALL"] = flag
        out[flag] = (timed(lambda: conv2d_wgrad_mfma(x, gy, None, None, 3, 1, 1, False)), conv2d_wgrad_mfma(x, gy, None, None, 3, 1, 1, False))
    err = float((out["0"][1] - out["1"][1]).abs().max() / out["0"][1].abs().max())
    print("B %d  %d -> %d  %3d^2   MFMA tiles %7.1f us   streaming %6.1f us   rel. difference %.1e" % (b, c, n, res, out["0"][0], out["1"][0], err), flush=True)
