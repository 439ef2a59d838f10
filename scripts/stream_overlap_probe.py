"""GPU box: do a matrix-bound kernel chain (weight gradients) and an HBM-bound chain (activation backward / row-dot
passes) overlap when they are issued on two HIP streams?  Serial on one stream vs concurrent, same process.
usage: python scripts/stream_overlap_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd.op import conv as cv  # noqa: E402
from stylerenderer_amd.op.fused_elem import rowdot  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
# weight-gradient chain: the 128^2 x 256 and 256^2 x 128 stride-1 layers of the headline step (Winograd wgrad)
xs = [torch.randn(16, 256, 128, 128, device=dev), torch.randn(16, 128, 256, 256, device=dev)]
gs = [torch.randn_like(x) for x in xs]
# HBM-bound chain: activation backward + row-dot on the 256^2 x 128 tensors
a, b = torch.randn(16, 128, 256, 256, device=dev), torch.randn(16, 128, 256, 256, device=dev)
s = torch.randn(16, 128, device=dev)
noise, nw, ab = torch.randn(16, 1, 256, 256, device=dev), torch.randn(1, device=dev), torch.randn(128, device=dev)


def mfma_chain():
    for _ in range(2):
        for x, g in zip(xs, gs):
            cv.conv2d_wgrad_mfma(x, g, None, None, 3, 1, 1, False)


def hbm_chain():
    for _ in range(4):
        cv._nba_bwd_dot(a, b, noise, nw, ab, 0.2, 2 ** 0.5, True)
        rowdot(a, b, s)


side = torch.cuda.Stream()


def serial():
    mfma_chain()
    hbm_chain()


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        mfma_chain()
    hbm_chain()
    main.wait_stream(side)


def timed(fn, k=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e3


tm, th = timed(mfma_chain), timed(hbm_chain)
print("matrix chain alone %.3f ms, HBM chain alone %.3f ms, sum %.3f" % (tm, th, tm + th))
for r in range(3):
    print("round %d: serial %.3f ms   two streams %.3f ms" % (r, timed(serial), timed(forked)))
