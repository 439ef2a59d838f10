"""TEST INFRASTRUCTURE — functional, CPU-only restatement of the reference generator's forward
pass, driven by a state_dict (no nn.Module tree).  It follows the REFERENCE's formulation, not
the product's: per-sample modulated weights and a grouped convolution.

  mapping network            reference model.py:145-146, layers.py:100-105, 222-248
  modulated convolution      reference layers.py:293-323   (F.conv2d / F.conv_transpose2d, groups = batch)
  noise + bias + LeakyReLU   reference model.py:26-32, layers.py:328-332, op/fused_act.py:87-94 (CPU branch)
  ToRGB + skip upsample      reference model.py:63-69, layers.py:170-181
  synthesis loop             reference model.py:172-182

PINNING: tests/test_oracle_model.py feeds it the same closed-form weights / latents / noise as
oracle/make_golden.py fed the reference and compares with tests/golden/generator_s8.npz and
generator_s64.npz (images from the reference itself).  The dense contraction is torch's conv2d —
an unpinned third-party dependency of the reference — hence a tolerance, not bits.

Used by tests (on-box oracle for the HIP generator at sizes the CPU finishes in seconds) and by
bench.py's cpu_baseline leg ("port": this code timed on the GPU box's host cores).
"""
import math

import torch
from torch.nn import functional as F


def _lrelu_bias(x, bias, scale=math.sqrt(2.0)):
    shape = [1, -1] + [1] * (x.dim() - 2)
    return F.leaky_relu(x + bias.view(*shape), 0.2) * scale


def _upfirdn(x, kernel, up=1, pad=(0, 0)):
    n, c, h, w = x.shape
    t = x.reshape(n * c, 1, h, w)
    if up > 1:
        z = t.new_zeros(n * c, 1, h * up, w * up)
        z[:, :, ::up, ::up] = t
        t = z
    t = F.pad(t, [pad[0], pad[1], pad[0], pad[1]])
    t = F.conv2d(t, torch.flip(kernel, [0, 1])[None, None])
    return t.reshape(n, c, t.shape[2], t.shape[3])


def mapping(sd, z, n_mlp, lr_mul=0.01):
    x = z * torch.rsqrt(torch.mean(z * z, -1, keepdim=True) + 1e-8)
    for i in range(1, n_mlp + 1):
        w, b = sd["style.%d.weight" % i], sd["style.%d.bias" % i]
        x = _lrelu_bias(F.linear(x, w * (lr_mul / math.sqrt(w.shape[1]))), b * lr_mul)
    return x


def modulated_conv(sd, prefix, x, style, demodulate=True, upsample=False):
    w = sd[prefix + ".weight"]                               # [1, Co, Ci, k, k]
    mw, mb = sd[prefix + ".modulation.weight"], sd[prefix + ".modulation.bias"]
    b, ci, h, wd = x.shape
    co, k = w.shape[1], w.shape[-1]
    s = F.linear(style, mw * (1.0 / math.sqrt(mw.shape[1])), bias=mb)
    wgt = (1.0 / math.sqrt(ci * k * k)) * w * s.view(b, 1, ci, 1, 1)
    if demodulate:
        wgt = wgt * torch.rsqrt(wgt.pow(2).sum([2, 3, 4], keepdim=True) + 1e-8)
    if upsample:
        wt = wgt.transpose(1, 2).reshape(b * ci, co, k, k)
        out = F.conv_transpose2d(x.reshape(1, b * ci, h, wd), wt, stride=2, groups=b)
        out = out.view(b, co, out.shape[2], out.shape[3])
        return _upfirdn(out, sd[prefix + ".blur.kernel"], pad=(1, 1))
    out = F.conv2d(x.reshape(1, b * ci, h, wd), wgt.view(b * co, ci, k, k), padding=k // 2, groups=b)
    return out.view(b, co, h, wd)


def styled_conv(sd, prefix, x, style, noise, upsample=False):
    out = modulated_conv(sd, prefix + ".conv", x, style, True, upsample)
    if noise is None:
        noise = torch.randn(out.shape[0], 1, out.shape[2], out.shape[3])
    out = out + sd[prefix + ".noise.weight"] * noise
    return _lrelu_bias(out, sd[prefix + ".activate.bias"])


def to_rgb(sd, prefix, x, style, skip=None):
    out = modulated_conv(sd, prefix + ".conv", x, style, demodulate=False) + sd[prefix + ".bias"]
    if skip is not None:
        out = out + _upfirdn(skip, sd[prefix + ".upsample.kernel"], up=2, pad=(2, 1))
    return out


def generator_forward(sd, size, z, noise=None, n_mlp=8):
    """sd: state_dict (tensors may require grad); z [B, D]; noise: list of num_layers tensors or None."""
    log_size = int(math.log(size, 2))
    num_layers = (log_size - 2) * 2 + 1
    if noise is None:
        noise = [None] * num_layers
    wl = mapping(sd, z, n_mlp)
    out = sd["input.input"].repeat(z.shape[0], 1, 1, 1)
    out = styled_conv(sd, "conv1", out, wl, noise[0])
    skip = to_rgb(sd, "to_rgb1", out, wl)
    for r in range(log_size - 2):
        out = styled_conv(sd, "convs.%d" % (2 * r), out, wl, noise[1 + 2 * r], upsample=True)
        out = styled_conv(sd, "convs.%d" % (2 * r + 1), out, wl, noise[2 + 2 * r])
        skip = to_rgb(sd, "to_rgbs.%d" % r, out, wl, skip)
    return skip


def time_generator_fwd_bwd(size=256, batch=2, iters=2, warmup=1, seed=0):
    """CPU baseline leg of bench.py: img/s of generator forward + backward on the host cores."""
    import time

    from stylerenderer_amd import model as product_model   # only for shapes / initial values

    torch.manual_seed(seed)
    g = product_model.Generator(size, 512, 8)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith("kernel")
                                              and not k.startswith("noises."))
          for k, v in g.state_dict().items()}
    leaves = [v for v in sd.values() if v.requires_grad]
    times = []
    for it in range(warmup + iters):
        z = torch.randn(batch, 512)
        t0 = time.perf_counter()
        img = generator_forward(sd, size, z)
        torch.autograd.grad(img.sum(), leaves, allow_unused=True)
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    return batch * len(times) / sum(times)
