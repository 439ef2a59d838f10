"""GPU box: k_conv_wino per-launch time against the batch size (inside a captured graph, weights U ready):
is there a fixed cost per launch at the training batch (4 / 8) that the benchmark batch (16) amortises?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylerenderer_amd import graphs  # noqa: E402
from stylerenderer_amd.op.conv import conv2d_mfma  # noqa: E402

dev = "cuda"
for (c, n, res) in ((128, 128, 256), (256, 256, 128), (512, 512, 64), (512, 512, 32)):
    line = []
    for b in (1, 2, 4, 8, 16):
        x = torch.randn(b, c, res, res, device=dev)
        wt = torch.randn(9, c, n, device=dev)
        isc, osc = torch.randn(b, c, device=dev), torch.randn(b, n, device=dev)

        def body():
            for _ in range(10):
                conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)

        body()
        torch.cuda.synchronize()
        g = graphs.capture(body)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        fl = 2.0 * b * res * res * c * n * 9
        line.append("B%d %.3f ms (%.0f TF alg, %.3f ms/img)" % (b, ms, fl / ms / 1e9, ms / b))
    print("%d->%d @%d^2 (incl. k_wino_weights): %s" % (c, n, res, "  ".join(line)), flush=True)
