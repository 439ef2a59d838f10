"""Blur (4x4 FIR, up = down = 1) after the up-sampling convolutions: time and HBM GB/s per shape (algorithmic bytes)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylerenderer_amd.op.upfirdn2d import upfirdn2d
dev = "cuda"
k1 = torch.tensor([1., 3., 3., 1.])
k = (k1[:, None] * k1[None, :]); k = (k / k.sum() * 4).to(dev)
for (b, c, res) in ((16, 128, 257), (16, 256, 129), (16, 512, 65), (16, 128, 255)):
    x = torch.randn(b, c, res, res, device=dev)
    pad = (1, 1) if res % 2 else (2, 2)
    for _ in range(3): y = upfirdn2d(x, k, pad=pad)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): y = upfirdn2d(x, k, pad=pad)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    byt = (x.numel() + y.numel()) * 4
    print("fir4 B%d C%d %d^2 -> %d^2: %.1f us  %.0f GB/s  checksum %.6e" % (b, c, res, y.shape[-1], dt * 1e6, byt / dt / 1e9, y.double().sum().item()), flush=True)
