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

# ---- backward of the up-sampling layer's tail: two kernels (k_nba_bwd<true> + k_fir4_tile<false>) vs k_fir4_nba_bwd
from stylerenderer_amd.op import conv as cv  # noqa: E402
from stylerenderer_amd.op.upfirdn2d import flipped, upfirdn2d_op  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / reps


for (b, c, oh) in ((16, 128, 256), (16, 256, 128), (16, 512, 64), (4, 128, 256)):
    gy, out = torch.randn(b, c, oh, oh, device=dev), torch.randn(b, c, oh, oh, device=dev)
    noise, nw, ab = torch.randn(b, 1, oh, oh, device=dev), torch.randn(1, device=dev), torch.randn(c, device=dev)
    shape257 = (b, c, oh + 1, oh + 1)
    kf = flipped(k)

    def two():
        gm, gb, gnw, rd = cv._nba_bwd_dot(gy, out, noise, nw, ab, 0.2, 2 ** 0.5, True)
        return upfirdn2d_op(gm.reshape(-1, oh, oh, 1), kf, 1, 1, 1, 1, 2, 2, 2, 2)

    def one():
        return cv._blur_nba_bwd(gy, out, kf, 1, shape257, noise, nw, ab, 0.2, 2 ** 0.5, True)[0]

    t2, t1 = timeit(two), timeit(one)
    alg = (2 * gy.numel() + (oh + 1) ** 2 * b * c) * 4
    print("tail bwd B%d C%d %d^2: two kernels %.1f us, one pass %.1f us (%.0f GB/s of its 3 tensors)  x%.2f" % (
        b, c, oh, t2 * 1e6, t1 * 1e6, alg / t1 / 1e9, t2 / t1), flush=True)
