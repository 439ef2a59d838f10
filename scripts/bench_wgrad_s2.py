#!/usr/bin/env python3
"""GPU box: stride-2 weight-gradient kernels at the generator's up-convolution shapes, DMA vs dword staging."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd.op.conv import conv2d_wgrad_mfma  # noqa: E402

dev = "cuda"


def run(b, c, n, g, tr, iters=10):
    out = 2 * g + 1 if tr else (g - 3) // 2 + 1
    x = torch.randn(b, c, g, g, device=dev)
    gy = torch.randn(b, n, out, out, device=dev)
    xs, gs = torch.randn(b, c, device=dev), torch.randn(b, n, device=dev)
    grid = g if tr else out
    fl = 2.0 * b * grid * grid * c * n * 9
    res = []
    for mode in ("1", "0"):
        os.environ["SR_WGRAD_DMA"] = mode
        for _ in range(3):
            conv2d_wgrad_mfma(x, gy, xs, gs, 3, 2, 0, tr)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(iters):
            conv2d_wgrad_mfma(x, gy, xs, gs, 3, 2, 0, tr)
        torch.cuda.synchronize()
        dt = (time.time() - t) / iters
        res.append("%s %.3f ms %.1f TF" % ("dma" if mode == "1" else "dword", dt * 1e3, fl / dt / 1e12))
    print("B%d C%d N%d grid%d %s: %s" % (b, c, n, grid, "convT" if tr else "conv", " | ".join(res)), flush=True)


for rep in range(2):
    run(16, 256, 128, 128, True)
    run(16, 512, 256, 64, True)
    run(16, 512, 512, 32, True)
    run(16, 512, 512, 16, True)
    run(8, 128, 256, 257, False)
    run(8, 256, 512, 129, False)
    run(4, 256, 128, 128, True)
