"""Sample images from a trained generator — the reference's generate.py (lines 14-75) on this package:
same flags, same checkpoint keys ('g_ema', else 'g'), same truncation trick, `%06d.png` outputs.

    python -m stylerenderer_amd.generate --size 256 --pics 20 --truncation 0.7 --ckpt checkpoint/550000.pt

torchvision is not required: the grid / normalisation of `utils.save_image(sample, nrow=1, normalize=True,
range=(-1, 1))` is restated in `save_image` (PIL when present, raw PPM otherwise).
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from . import checkpoint


def to_uint8_grid(images, nrow=1, value_range=(-1, 1), padding=2):
    """torchvision.utils.make_grid + normalize semantics for [B, 3, H, W]: min-max to `value_range`, `nrow` images per
    row, `padding` pixels of zeros around every image."""
    lo, hi = value_range
    x = ((images.detach().float().cpu().clamp(lo, hi) - lo) / max(hi - lo, 1e-5))
    b, c, h, w = x.shape
    if b == 1:
        # make_grid returns a single image unpadded (`if tensor.size(0) == 1: return tensor.squeeze(0)`): the
        # reference's default --sample 1 writes size x size files
        return (x[0].mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)).numpy()
    xmaps = min(nrow, b)
    ymaps = int(math.ceil(b / xmaps))
    grid = torch.zeros(c, ymaps * (h + padding) + padding, xmaps * (w + padding) + padding)
    for i in range(b):
        r, col = divmod(i, xmaps)
        grid[:, padding + r * (h + padding): padding + r * (h + padding) + h,
             padding + col * (w + padding): padding + col * (w + padding) + w] = x[i]
    return (grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)).numpy()


def save_image(images, path, nrow=1, value_range=(-1, 1)):
    arr = to_uint8_grid(images, nrow, value_range)
    try:
        from PIL import Image

        Image.fromarray(arr).save(path)
    except ImportError:
        with open(os.path.splitext(path)[0] + ".ppm", "wb") as f:
            f.write(b"P6 %d %d 255\n" % (arr.shape[1], arr.shape[0]) + np.ascontiguousarray(arr).tobytes())


def generate(args, g_ema, mean_latent, folder="sample"):
    bits = max(int(math.floor(math.log(max(args.pics, 1)) / math.log(10))) + 1, 6)
    fmt = "%%0%dd.png" % bits
    os.makedirs(folder, exist_ok=True)
    with torch.no_grad():
        g_ema.eval()
        for i in range(args.pics):
            sample_z = torch.randn(args.sample, args.latent, device=args.device)
            sample, _ = g_ema([sample_z], truncation=args.truncation, truncation_latent=mean_latent)
            save_image(sample, os.path.join(folder, fmt % i), nrow=1, value_range=(-1, 1))
    return args.pics


def main(argv=None):
    ap = argparse.ArgumentParser(description="Generate samples from the generator")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--sample", type=int, default=1)
    ap.add_argument("--pics", type=int, default=20)
    ap.add_argument("--truncation", type=float, default=1)
    ap.add_argument("--truncation_mean", type=int, default=4096)
    ap.add_argument("--ckpt", type=str, default="stylegan2-ffhq-config-f.pt")
    ap.add_argument("--output", type=str, default="sample")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--seed", type=int, default=-1)
    ap.add_argument("--channel_multiplier", type=int, default=2)
    args = ap.parse_args(argv)
    if args.seed < 0:
        args.seed = int(time.time())
    torch.manual_seed(args.seed)
    if torch.cuda.is_available() and 0 <= args.gpu < torch.cuda.device_count():
        args.device = "cuda:%d" % args.gpu
    else:
        args.device = "cpu"
    args.latent, args.n_mlp = 512, 8
    g_ema = checkpoint.load_generator(args.ckpt, args.size, args.latent, args.n_mlp, args.channel_multiplier,
                                      device=args.device)
    mean_latent = None
    if args.truncation < 1:
        with torch.no_grad():
            mean_latent = g_ema.mean_latent(args.truncation_mean)
    return generate(args, g_ema, mean_latent, args.output)


if __name__ == "__main__":
    main()
