"""Transposed (upsampling) convolution: fused four-phase kernel vs per-phase launches (SR_CONVT_FUSED=0)."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from stylerenderer_amd.op.conv import conv2d_mfma

    tag = "fused " if os.environ.get("SR_CONVT_FUSED", "1") != "0" else "phases"
    for (b, c, n, res) in ((16, 256, 128, 128), (16, 512, 256, 64), (16, 512, 512, 32)):
        x = torch.randn(b, c, res, res, device="cuda")
        wt = torch.randn(9, c, n, device="cuda")
        isc = torch.randn(b, c, device="cuda")
        osc = torch.randn(b, n, device="cuda")
        for _ in range(2):
            y = conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, True)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(20):
            y = conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, True)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 20
        print("%s convT B%d C%d N%d res%d: %.3f ms  %.1f TFLOP/s" % (tag, b, c, n, res, dt * 1e3,
                                                                    2.0 * b * res * res * c * n * 9 / dt / 1e12), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for mode in ("1", "0"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SR_CONVT_FUSED=mode),
                           check=False, timeout=600)
