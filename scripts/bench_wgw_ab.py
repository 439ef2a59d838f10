#!/usr/bin/env python3
"""GPU box: Winograd weight-gradient kernels, two waves per SIMD (k_wgrad_wino2) vs one (k_wgrad_wino)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd.op.conv import conv2d_wgrad_mfma  # noqa: E402

dev = "cuda"
for rep in range(2):
    for (b, c, n, res) in ((16, 128, 128, 256), (16, 256, 256, 128), (16, 512, 512, 64), (16, 512, 512, 32),
                           (4, 128, 128, 256), (8, 256, 256, 128)):
        x = torch.randn(b, c, res, res, device=dev)
        gy = torch.randn(b, n, res, res, device=dev)
        xs, gs = torch.randn(b, c, device=dev), torch.randn(b, n, device=dev)
        fl = 2.0 * b * res * res * c * n * 9
        out = []
        for mode in ("1", "0"):
            os.environ["SR_WGW_WAVES"] = "8" if mode == "1" else "4"
            for _ in range(3):
                conv2d_wgrad_mfma(x, gy, xs, gs, 3, 1, 1)
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(10):
                conv2d_wgrad_mfma(x, gy, xs, gs, 3, 1, 1)
            torch.cuda.synchronize()
            dt = (time.time() - t) / 10
            out.append("%s %.3f ms %.1f TF (%.2f executed)" % ("v2" if mode == "1" else "v1", dt * 1e3, fl / dt / 1e12,
                                                               fl / dt / 1e12 * 16 / 36 / 157.3))
        print("B%d C%d N%d res%d: %s" % (b, c, n, res, " | ".join(out)), flush=True)
