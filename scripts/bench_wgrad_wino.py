"""A/B of the Winograd weight-gradient kernel against the direct one (SR_WINOGRAD=0) + error vs float64."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from torch.nn import functional as F
    from stylerenderer_amd.op.conv import conv2d_wgrad_mfma

    dev = "cuda"
    tag = "wino " if os.environ.get("SR_WINOGRAD", "1") != "0" else "direct"
    g = torch.Generator().manual_seed(3)
    b, c, n, h, w = 3, 64, 128, 8, 32
    x = torch.randn(b, c, h, w, generator=g)
    gy = torch.randn(b, n, h, w, generator=g)
    xs = torch.randn(b, c, generator=g)
    gs = torch.randn(b, n, generator=g)
    wref = torch.zeros(n, c, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double() * xs.double()[:, :, None, None], wref, padding=1)
    (gw,) = torch.autograd.grad(y, wref, gy.double() * gs.double()[:, :, None, None])
    ymag = F.conv2d(x.abs().double() * xs.abs().double()[:, :, None, None], wref, padding=1)
    (mag,) = torch.autograd.grad(ymag, wref, gy.abs().double() * gs.abs().double()[:, :, None, None])
    want = gw.permute(2, 3, 1, 0).reshape(9, c, n)
    mag = mag.permute(2, 3, 1, 0).reshape(9, c, n)
    got = conv2d_wgrad_mfma(x.to(dev), gy.to(dev), xs.to(dev), gs.to(dev), 3, 1, 1).cpu().double()
    print("%s small-shape max |err| / sum|a*b| = %.3e  max|err| %.3e  max|ref| %.3e" % (
        tag, ((got - want).abs() / mag).max().item(), (got - want).abs().max().item(), want.abs().max().item()),
        flush=True)
    for (b, c, n, res) in ((16, 128, 128, 256), (16, 256, 256, 128), (16, 512, 512, 64), (16, 512, 512, 32),
                           (16, 512, 512, 16)):
        x = torch.randn(b, c, res, res, device=dev)
        gy = torch.randn(b, n, res, res, device=dev)
        xs = torch.randn(b, c, device=dev)
        gs = torch.randn(b, n, device=dev)
        for _ in range(2):
            y = conv2d_wgrad_mfma(x, gy, xs, gs, 3, 1, 1)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(5):
            y = conv2d_wgrad_mfma(x, gy, xs, gs, 3, 1, 1)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 5
        fl = 2.0 * b * res * res * c * n * 9
        print("%s WGRAD B%d C%d N%d res%d: %.3f ms  %.1f TFLOP/s (direct FLOPs)" % (tag, b, c, n, res, dt * 1e3,
                                                                                    fl / dt / 1e12), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for mode in ("1", "0"):
            env = dict(os.environ, SR_WINOGRAD=mode)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False, timeout=600)
