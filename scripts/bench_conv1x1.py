"""1x1 convolutions of the discriminator (ResBlock.skip after the decimating blur, from-RGB) and their data / weight
gradients, timed inside a captured graph; the strided-batched library GEMM of the same contraction beside them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylerenderer_amd import graphs  # noqa: E402
from stylerenderer_amd.op.conv import conv2d_mfma, conv2d_wgrad_mfma  # noqa: E402


def timed(body, reps=10):
    def many():
        for _ in range(reps):
            body()
    many()
    torch.cuda.synchronize()
    g = graphs.capture(many)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps)


for (b, c, n, res) in ((8, 64, 128, 128), (8, 128, 256, 64), (8, 256, 512, 32), (8, 512, 512, 16), (4, 128, 256, 64),
                       (8, 128, 64, 128), (8, 256, 128, 64)):
    x = torch.randn(b, c, res, res, device="cuda")
    wt = torch.randn(1, c, n, device="cuda")
    gy = torch.randn(b, n, res, res, device="cuda")
    w2 = wt[0].t().contiguous()                                   # [n, c]
    flops = 2.0 * b * res * res * c * n
    byt = 4.0 * b * res * res * (c + n)
    t_conv = timed(lambda: conv2d_mfma(x, wt, None, None, None, 1, 1, 0, False))
    t_wg = timed(lambda: conv2d_wgrad_mfma(x, gy, None, None, 1, 1, 0, False))
    t_lib = timed(lambda: torch.matmul(w2, x.view(b, c, -1)))
    print("1x1 B%d %d->%d @%d^2: conv %.1f us (%.0f TF, %.2f TB/s)  wgrad %.1f us (%.0f TF)  library bmm %.1f us   floors: mfma %.1f us, hbm %.1f us"
          % (b, c, n, res, t_conv * 1e3, flops / t_conv / 1e9, byt / t_conv / 1e9, t_wg * 1e3, flops / t_wg / 1e9,
             t_lib * 1e3, flops / 157.3e12 * 1e6, byt / 8e12 * 1e6), flush=True)
