"""A/B of the Winograd F(2x2,3x3) path against the direct implicit GEMM (SR_WINOGRAD=0), plus the
error of both against a float64 reference.  Run on the GPU box:  python scripts/bench_wino.py"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from torch.nn import functional as F
    from stylerenderer_amd.op.conv import conv2d_mfma

    dev = "cuda"
    tag = "wino " if os.environ.get("SR_WINOGRAD", "1") != "0" else "direct"
    # accuracy on a small eligible shape
    g = torch.Generator().manual_seed(1)
    b, c, n, h, w = 2, 24, 64, 16, 32
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(n, c, 3, 3, generator=g) / (3 * c ** 0.5)
    isc = torch.randn(b, c, generator=g)
    osc = torch.randn(b, n, generator=g)
    bias = torch.randn(n, generator=g)
    wt = wgt.permute(2, 3, 1, 0).reshape(9, c, n).contiguous()
    ref = F.conv2d(x.double() * isc.double()[:, :, None, None], wgt.double(), padding=1)
    ref = ref * osc.double()[:, :, None, None] + bias.double()[None, :, None, None]
    mag = F.conv2d(x.abs().double() * isc.abs().double()[:, :, None, None], wgt.abs().double(), padding=1)
    mag = mag * osc.abs().double()[:, :, None, None]
    got = conv2d_mfma(x.to(dev), wt.to(dev), isc.to(dev), osc.to(dev), bias.to(dev), 3, 1, 1).cpu().double()
    err = ((got - ref).abs() / mag).max().item()
    print("%s small-shape max |err| / sum|a*b| = %.3e   max|err| = %.3e" % (tag, err, (got - ref).abs().max().item()),
          flush=True)
    for (b, c, n, res) in ((16, 128, 128, 256), (16, 256, 256, 128), (16, 512, 512, 64), (16, 512, 512, 32)):
        x = torch.randn(b, c, res, res, device=dev)
        wt = torch.randn(9, c, n, device=dev)
        isc = torch.randn(b, c, device=dev)
        osc = torch.randn(b, n, device=dev)
        for _ in range(2):
            y = conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(20):
            y = conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 20
        fl = 2.0 * b * res * res * c * n * 9
        print("%s B%d C%d N%d res%d: %.3f ms  %.1f TFLOP/s (direct-conv FLOPs)" % (tag, b, c, n, res, dt * 1e3,
                                                                                   fl / dt / 1e12), flush=True)
    print("finite:", bool(torch.isfinite(y).all()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for mode in ("1", "0"):
            env = dict(os.environ, SR_WINOGRAD=mode)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False, timeout=600)
