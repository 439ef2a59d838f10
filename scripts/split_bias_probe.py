"""Mean SIGNED error of the split-bf16 stride-2 convolution against float64 on all-positive operands (where a systematic
component cannot hide behind cancelling signs), next to the exact-fp32 MFMA kernel."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from stylerenderer_amd.op.conv import conv2d_mfma  # noqa: E402
from test_conv_gpu import to_taps  # noqa: E402

DEV = "cuda"
for (b, c, n, res, tr) in ((1, 256, 512, 129, False), (1, 512, 512, 65, False), (2, 64, 128, 65, False),
                           (1, 512, 256, 64, True), (1, 256, 128, 128, True)):
    g = torch.Generator().manual_seed(5)
    for positive in (False, True):
        x = torch.randn(b, c, res, res, generator=g)
        wgt = torch.randn((c, n, 3, 3) if tr else (n, c, 3, 3), generator=g)
        if positive:
            x, wgt = x.abs(), wgt.abs()
        conv = F.conv_transpose2d if tr else F.conv2d
        want = conv(x.double(), wgt.double(), stride=2)
        mag = conv(x.abs().double(), wgt.abs().double(), stride=2)
        out = {}
        for mode in ("0", "1"):
            os.environ["SR_CONV_SPLIT_BF16"] = mode
            out[mode] = conv2d_mfma(x.to(DEV), to_taps(wgt, tr).to(DEV), None, None, None, 3, 2, 0, tr).cpu().double()
        row = []
        for mode in ("0", "1"):
            e = (out[mode] - want) / mag
            row.append("%s: max %.2e mean %+.2e rms %.2e" % ("exact" if mode == "0" else "split", e.abs().max(), e.mean(),
                                                           e.pow(2).mean().sqrt()))
        print((b, c, n, res, "T" if tr else "C"), "positive" if positive else "signed  ", " | ".join(row))
