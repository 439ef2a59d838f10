"""Networks of the StyleRenderer hot path on the MI355X-native layers.

Drop-in for the reference's model.py: same class names, constructor signatures, forward
signatures and return tuples, and the same state_dict keys (incl. the reference's duplicated
`to_rgbs` tail, SURVEY.md D5: the second half of `to_rgbs` is registered but never used, so
reference checkpoints with 165 keys at 256x256 load with strict=True).

  Generator         reference model.py:71-187
  GeneratorWithMap  reference model.py:188-295   (rasterised normal maps modulate every layer)
  Discriminator     reference model.py:296-336
"""
import math

import numpy as np
import torch
from torch import nn

from .layers import (BiasBank, Blur, ConstantInput, ConvLayer, EqualLinear, ModulatedConv2d, NoiseInjection,  # noqa: F401
                     PixelNorm, ResBlock, Upsample)
from .op import FusedLeakyReLU, rasterize
from .op import smallconv as _smallconv
from .op import style_bank as _style_bank
from .op import weight_bank as _weight_bank
from .op.fused_elem import blur_noise_bias_act, noise_bias_act, noise_bias_act_affine
from .op.rasterize import rasterize_pyramid
from .op.upfirdn2d import upsample2_add

CHANNEL_BASE = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128, 256: 64, 512: 32, 1024: 16}


def channel_table(channel_multiplier):
    return {r: (c if r <= 32 else c * channel_multiplier) for r, c in CHANNEL_BASE.items()}


class StyledConv(nn.Module):
    """modulated conv -> noise -> bias + LeakyReLU (reference model.py:11-32)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False,
                 blur_kernel=[1, 3, 3, 1], demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim,
                                    upsample=upsample, blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.bias = None
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None):
        if noise is not None and noise.requires_grad:
            # optimising the noise maps (projector-style use): the fused nodes do not differentiate with
            # respect to `noise`, the separate operators do (reference layers.py:324-332)
            return self.activate(self.noise(self.conv(input, style), noise=noise))
        if input.device.type == "cuda" and self.conv.upsample:
            # upsampling layer: the blur after the transposed conv, the noise injection, the bias and the
            # LeakyReLU are one kernel (one pass over the activation instead of three)
            blur = self.conv.blur
            oh = 2 * input.shape[2] + 1 + sum(blur.pad) - 3
            ow = 2 * input.shape[3] + 1 + sum(blur.pad) - 3
            if noise is None:
                noise = input.new_empty(input.shape[0], 1, oh, ow).normal_()
            noise = noise.contiguous()
            fused = self.conv.forward_up_noise_bias_act(input, style, noise, self.noise.weight, self.activate.bias,
                                                        self.activate.negative_slope, self.activate.scale)
            if fused is not None:
                return fused
            y = self.conv(input, style, skip_blur=True)
            return blur_noise_bias_act(y, blur.kernel, blur.pad, noise, self.noise.weight, self.activate.bias,
                                       self.activate.negative_slope, self.activate.scale)
        if input.device.type == "cuda":
            if noise is None:
                noise = input.new_empty(input.shape[0], 1, input.shape[2], input.shape[3]).normal_()
            # 32^2 .. 256^2 layers: the Winograd convolution applies noise + bias + LeakyReLU in its store
            fused = self.conv.forward_noise_bias_act(input, style, noise.contiguous(), self.noise.weight,
                                                     self.activate.bias, self.activate.negative_slope,
                                                     self.activate.scale)
            if fused is not None:
                return fused
            out = self.conv(input, style)
            # noise injection + bias + LeakyReLU in one pass over the activation
            return noise_bias_act(out, noise, self.noise.weight, self.activate.bias,
                                  self.activate.negative_slope, self.activate.scale)
        out = self.conv(input, style)
        out = self.noise(out, noise=noise)
        return self.activate(out)


class StyledMapConv(nn.Module):
    """StyledConv with a per-pixel affine from the rasterised normal map between the conv and
    the noise (reference model.py:33-55)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False,
                 blur_kernel=[1, 3, 3, 1], demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim,
                                    upsample=upsample, blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.bias = None
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, stylemap, noise=None):
        out = self.conv(input, style)
        if out.device.type == "cuda" and not (noise is not None and noise.requires_grad):
            if noise is None:
                noise = out.new_empty(out.shape[0], 1, out.shape[2], out.shape[3]).normal_()
            # per-pixel affine + noise + bias + LeakyReLU in one pass over the activation (and one in backward)
            sm = stylemap if stylemap.shape[1] == 2 else stylemap[:, :2]
            return noise_bias_act_affine(out, sm, noise, self.noise.weight, self.activate.bias,
                                         self.activate.negative_slope, self.activate.scale)
        out = out * stylemap[:, :1] + stylemap[:, 1:2]
        out = self.noise(out, noise=noise)
        return self.activate(out)


def _fork_enabled():
    """SR_TORGB_FORK=0: ToRGB as a second consumer of the feature map (A/B)."""
    import os

    return os.environ.get("SR_TORGB_FORK", "1") != "0"


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None, fork=False):
        """fork (the synthesis loops): returns (input', rgb) with input' = input for the NEXT layer to consume — on device
        tensors the feature map then has one consumer node, whose backward adds the two gradients inside the
        data-gradient kernel (op.smallconv.SmallConvFork); elsewhere input' is input."""
        conv = self.conv
        if _smallconv.supported(input, conv.out_channel):
            # device tensors: the bias rides in the streaming 1x1 kernel and the skip addition in the up-sampling
            # kernel's store — two launches for conv + bias + upsample + add (reference model.py:63-69)
            out = _smallconv.modulated_conv1x1_small(input, conv.weight.view(conv.out_channel, conv.in_channel),
                                                     conv.style_of(style), self.bias.view(-1), scale=conv.scale,
                                                     fork=fork and _fork_enabled())
            if fork and _fork_enabled():
                input, out = out
            if skip is not None:
                out = upsample2_add(skip, self.upsample.kernel, self.upsample.pad, out)
            return (input, out) if fork else out
        out = self.conv(input, style) + self.bias
        if skip is not None:
            out = out + self.upsample(skip)
        return (input, out) if fork else out


class Generator(nn.Module):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1],
                 lr_mlp=0.01):
        super().__init__()
        self._build_trunk(size, style_dim, n_mlp, channel_multiplier, lr_mlp)
        self._build_synthesis(StyledConv, style_dim, blur_kernel)

    # -- registration order matters for state_dict key order; names match the reference
    def _build_trunk(self, size, style_dim, n_mlp, channel_multiplier, lr_mlp):
        self.size = size
        self.style_dim = style_dim
        mapping = [PixelNorm()]
        mapping += [EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu")
                    for _ in range(n_mlp)]
        self.style = nn.Sequential(*mapping)
        self.channels = channel_table(channel_multiplier)
        self.input = ConstantInput(self.channels[4])
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer("noise_%d" % layer_idx, torch.randn(1, 1, 2 ** res, 2 ** res))
        # the ToRGB heads the forward pass uses (first half of `to_rgbs`)
        for i in range(3, self.log_size + 1):
            self.to_rgbs.append(ToRGB(self.channels[2 ** i], style_dim))

    def _build_synthesis(self, conv_cls, style_dim, blur_kernel, per_resolution=None):
        in_channel = self.channels[4]
        self.conv1 = conv_cls(in_channel, in_channel, 3, style_dim, blur_kernel=blur_kernel)
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(conv_cls(in_channel, out_channel, 3, style_dim, upsample=True,
                                       blur_kernel=blur_kernel))
            self.convs.append(conv_cls(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            if per_resolution is not None:
                per_resolution()
            # never used by forward(); kept so that reference checkpoints load
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel

    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            noises += [torch.randn(1, 1, 2 ** i, 2 ** i, device=device) for _ in range(2)]
        return noises

    def mean_latent(self, n_latent):
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(z).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    # -- shared front half of forward(): styles -> per-layer latent [B, n_latent, D] and noises
    def _latents(self, styles, inject_index, truncation, truncation_latent, input_is_latent, noise,
                 randomize_noise):
        if not input_is_latent:
            if (len(styles) > 1 and styles[0].device.type == "cuda"
                    and all(s.shape == styles[0].shape for s in styles[1:])):
                # style mixing on the device: the mapping network (row-wise: PixelNorm + 8 EqualLinear) runs ONCE
                # over the stacked latents — half the launches of its forward and backward, and every weight
                # receives one gradient instead of two that autograd would have to add
                styles = list(self.style(torch.cat(list(styles), 0)).chunk(len(styles), 0))
            else:
                styles = [self.style(s) for s in styles]
        if noise is None:
            if randomize_noise:
                noise = self._draw_noise(styles[0])
            else:
                noise = [getattr(self.noises, "noise_%d" % i) for i in range(self.num_layers)]
        if truncation < 1 and truncation_latent is not None:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if len(styles) < 2:
            inject_index = self.n_latent
            if styles[0].dim() < 3:
                latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1)
            else:
                latent = styles[0]
        else:
            if inject_index is None:
                inject_index = np.random.choice(self.n_latent - 2) + 1
            if torch.is_tensor(inject_index):
                # crossover point held on the device (captured training step: the same graph serves every
                # crossover, incl. inject_index == n_latent, i.e. no mixing): same latent as the cat below
                layer = torch.arange(self.n_latent, device=styles[0].device).view(1, -1, 1)
                latent = torch.where(layer < inject_index.view(1, 1, 1), styles[0].unsqueeze(1),
                                     styles[1].unsqueeze(1))
            else:
                latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                    styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        return latent, noise

    def _draw_noise(self, like):
        """Fresh per-sample noise maps for every layer (reference layers.py:328-332 draws one per NoiseInjection call):
        on a device ONE normal_() launch over a flat buffer, the maps are views of it (layer i: [B, 1, 2^r, 2^r] with
        r = (i + 5) // 2, i.e. 4, 8, 8, 16, 16, ...) — 13 launches less per forward at 256^2.  SR_NOISE_BANK=0, CPU
        tensors, or a resolution the module was not built for: None per layer (each layer draws its own)."""
        import os

        if like.device.type != "cuda" or os.environ.get("SR_NOISE_BANK", "1") == "0":
            return [None] * self.num_layers
        # the map sizes come from the module itself — the registered noise buffers [1, 1, h, w] have exactly the shapes
        # of the feature maps their layers see when the constant input is the 4 x 4 the module was built with; anything
        # else (a replaced constant input, missing buffers) falls back to the per-layer draw, which sizes the noise from
        # the image like the reference's NoiseInjection
        const = self.input.input
        shapes = [tuple(getattr(self.noises, "noise_%d" % i, torch.empty(0)).shape[-2:]) for i in range(self.num_layers)]
        if (tuple(const.shape[-2:]) != (4, 4) or const.dtype != torch.float32 or any(len(sh) != 2 for sh in shapes)
                or shapes[0] != (4, 4)):
            return [None] * self.num_layers
        b = like.shape[0]
        flat = torch.empty(b * sum(h * w for h, w in shapes), device=like.device, dtype=torch.float32).normal_()
        out, off = [], 0
        for h, w in shapes:
            out.append(flat[off:off + b * h * w].view(b, 1, h, w))
            off += b * h * w
        return out

    def _style_layers(self):
        """[(ModulatedConv2d, latent index)] in the order forward() calls them (reference model.py:172-186: conv1 0,
        to_rgb1 1, then per resolution conv_up i, conv i + 1, to_rgb i + 2 with i = 1, 3, 5, ...)."""
        seq = [(self.conv1.conv, 0), (self.to_rgb1.conv, 1)]
        i = 1
        for conv_up, conv, to_rgb in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            seq += [(conv_up.conv, i), (conv.conv, i + 1), (to_rgb.conv, i + 2)]
            i += 2
        return seq

    def _layer_styles(self, latent):
        """What each modulated layer receives as `style`, in call order: its latent row, or — device tensors — a
        StylePack from the batched evaluation of every modulation / demodulation of the pass (op.style_bank)."""
        seq = self._style_layers()
        if latent.device.type == "cuda" and latent.dtype == torch.float32 and _style_bank.enabled():
            return _style_bank.build(seq, latent)
        # one unbind (its backward is one stack) instead of a select per layer, whose backward would
        # materialise and add a zero-filled [B, n_latent, D] tensor 2 * n_latent times
        rows = latent.unbind(1)
        return [rows[li] for _, li in seq]

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True):
        latent, noise = self._latents(styles, inject_index, truncation, truncation_latent,
                                      input_is_latent, noise, randomize_noise)
        with _weight_bank.Scope(self):          # device tensors: every convolution weight prepared up front
            out = self.input(latent)
            st = self._layer_styles(latent)
            out = self.conv1(out, st[0], noise=noise[0])
            out, skip = self.to_rgb1(out, st[1], fork=True)      # `out` goes on to the next layer through the ToRGB node
            k = 2
            for conv_up, conv, n_up, n_conv, to_rgb in zip(self.convs[::2], self.convs[1::2],
                                                           noise[1::2], noise[2::2], self.to_rgbs):
                out = conv_up(out, st[k], noise=n_up)
                out = conv(out, st[k + 1], noise=n_conv)
                out, skip = to_rgb(out, st[k + 2], skip, fork=True)
                k += 3
        return skip, (latent if return_latents else None)


class GeneratorWithMap(Generator):
    def __init__(self, size, style_dim, n_mlp, n_stylemap=3, channel_multiplier=2,
                 blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        nn.Module.__init__(self)
        self._build_trunk(size, style_dim, n_mlp, channel_multiplier, lr_mlp)
        self.norm_to_style = nn.ModuleList()
        if n_stylemap != 3:
            self.norm1 = nn.Sequential(ConvLayer(3, n_stylemap, 3), ResBlock(n_stylemap, 2, downsample=False))
        else:
            self.norm1 = ResBlock(n_stylemap, 2, downsample=False)

        def add_map_heads():
            if n_stylemap != 3:
                self.norm_to_style.append(ConvLayer(3, n_stylemap, 3))
            self.norm_to_style.append(ResBlock(n_stylemap, 4, downsample=False))

        self._build_synthesis(StyledMapConv, style_dim, blur_kernel, per_resolution=add_map_heads)

    def forward(self, styles, mesh, return_normals=False, return_latents=False, inject_index=None,
                truncation=1, truncation_latent=None, input_is_latent=False, noise=None,
                randomize_noise=True):
        latent, noise = self._latents(styles, inject_index, truncation, truncation_latent,
                                      input_is_latent, noise, randomize_noise)
        vert, attr, tri = mesh[0], mesh[1], mesh[2]
        # device tensors: every convolution weight and every merged layer bias prepared up front
        with _weight_bank.Scope(self), BiasBank(self):
            return self._synthesis_with_maps(latent, noise, vert, attr, tri, return_normals, return_latents)

    def _synthesis_with_maps(self, latent, noise, vert, attr, tri, return_normals, return_latents):
        out = self.input(latent)
        # the reference hands the permuted VIEW of the rasterizer output on (model.py:262) — that is what `norm_maps`
        # returns and what the path-length regulariser differentiates against.  Here the rasterizer writes the maps
        # channel-major itself (SR_RASTER_CHW: same values, no re-layout pass forward or backward per resolution;
        # SR_RASTER_NCHW=0 keeps the permuted view + one contiguous copy for the map heads)
        res0 = (int(out.shape[2]), int(out.shape[3]))
        # device tensors: the normal maps of all resolutions from ONE node, whose backward adds the per-resolution mesh
        # gradients inside the gather kernels (op.rasterize.RasterizePyramid) instead of 2 tensor additions per resolution
        # (levels = what the loop below consumes: `to_rgbs` carries the reference's unused second half)
        pyramid = _normal_pyramid(vert, attr, tri, [(res0[0] << k, res0[1] << k)
                                                     for k in range(min(len(self.convs) // 2, len(self.to_rgbs)) + 1)])
        norm_maps = [pyramid[0] if pyramid else _normal_map(vert, attr, tri, res0[0], res0[1])]
        maps = self.norm1(norm_maps[-1].contiguous())
        st = self._layer_styles(latent)
        out = self.conv1(out, st[0], maps, noise=noise[0])
        out, skip = self.to_rgb1(out, st[1], fork=True)          # `out` goes on to the next layer through the ToRGB node
        two_stage = len(self.convs) == len(self.norm_to_style)
        i, k = 1, 2
        for conv_up, conv, n_up, n_conv, to_rgb in zip(self.convs[::2], self.convs[1::2],
                                                       noise[1::2], noise[2::2], self.to_rgbs):
            norm_maps.append(pyramid[len(norm_maps)] if pyramid else
                             _normal_map(vert, attr, tri, 2 * int(out.shape[2]), 2 * int(out.shape[3])))
            nm = norm_maps[-1].contiguous()
            if two_stage:
                maps = self.norm_to_style[i](self.norm_to_style[i - 1](nm))
            else:
                maps = self.norm_to_style[i // 2](nm)
            # one split (its backward is one cat) instead of two slices, whose backward zero-fills and adds two
            # full-size copies of `maps` — per resolution and per order of differentiation
            maps_up, maps_conv = maps.split([2, maps.shape[1] - 2], 1)
            out = conv_up(out, st[k], maps_up, noise=n_up)
            out = conv(out, st[k + 1], maps_conv, noise=n_conv)
            out, skip = to_rgb(out, st[k + 2], skip, fork=True)
            i += 2
            k += 3
        return skip, (latent if return_latents else None), (norm_maps if return_normals else None)


def _normal_pyramid(vert, attr, tri, sizes):
    """The channel-major normal maps of all `sizes` from one autograd node, or None (CPU tensors, the permuted-view layout
    SR_RASTER_NCHW=0, SR_RASTER_PYRAMID=0: the caller rasterises per resolution)."""
    import os

    if (vert.device.type != "cuda" or vert.dim() != 3 or attr.dim() != 3 or os.environ.get("SR_RASTER_NCHW", "1") == "0"
            or os.environ.get("SR_RASTER_PYRAMID", "1") == "0"):
        return None
    return rasterize_pyramid(vert, attr, tri, sizes, channel_major=True)


def _normal_map(vert, attr, tri, h, w):
    """Rasterised vertex attributes as [b, c, h, w] (reference model.py:262: rasterize(...).permute(0, 3, 1, 2))."""
    import os

    if os.environ.get("SR_RASTER_NCHW", "1") != "0":
        return rasterize(vert, attr, tri, h, w, channel_major=True)
    return rasterize(vert, attr, tri, h, w).permute(0, 3, 1, 2)


class Discriminator(nn.Module):
    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        channels = channel_table(channel_multiplier)
        convs = [ConvLayer(3, channels[size], 1)]
        log_size = int(math.log(size, 2))
        in_channel = channels[size]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.convs = nn.Sequential(*convs)
        self.stddev_group = 4
        self.stddev_feat = 1
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(
            EqualLinear(channels[4] * 4 * 4, channels[4], activation="fused_lrelu"),
            EqualLinear(channels[4], 1))

    def forward(self, input):
        # device tensors: every convolution weight and every merged layer bias prepared up front
        with _weight_bank.Scope(self), BiasBank(self):
            return self._forward(input)

    def _forward(self, input):
        out = self.convs(input)
        batch, channel, height, width = out.shape
        group = min(batch, self.stddev_group)
        # minibatch standard deviation over groups of `group` samples
        stat = out.view(group, -1, self.stddev_feat, channel // self.stddev_feat, height, width)
        stat = torch.sqrt(stat.var(0, unbiased=False) + 1e-8)
        stat = stat.mean([2, 3, 4], keepdim=True).squeeze(2).repeat(group, 1, height, width)
        out = self.final_conv(torch.cat([out, stat], 1))
        return self.final_linear(out.view(batch, -1))
