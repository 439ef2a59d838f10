"""Layers of the StyleRenderer generator / discriminator on the MI355X-native operators.

Same class names, constructor arguments, parameter / buffer names and shapes as the reference's
layers.py (so reference checkpoints load unchanged), different execution:

  * ModulatedConv2d (reference layers.py:259-323) never materialises per-sample weights.  The style
    scales the activation tile inside the MFMA convolution kernel, the demodulation factor
    rsqrt(sum (W*s)^2 + eps) = rsqrt(s^2 @ sum_k W^2 + eps) is a [B,Cin]x[Cin,Cout] product, and it
    is applied in the kernel epilogue.  Identical algebra, fp32 round-off level differences
    (tolerance stated in tests/test_model_gpu.py).
  * the stride-2 transposed convolution of the upsampling layers runs as four output phases of
    the same kernel, followed by the LDS-tiled FIR blur (op.upfirdn2d).
  * CPU tensors follow the reference's own formulation (grouped F.conv2d), which is what the
    reference executes for them; device tensors never take that route.
"""
import math
import os

import torch
from torch import nn
from torch.nn import functional as F

from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d
from .op import conv as _conv
from .op import conv_generic as _convg
from .op import smallconv as _smallconv
from .op import style as _style
from .op.style_bank import StylePack
from .op.upfirdn2d import skip_down as _skip_down
from .op.weight_prep import weight_prep as _weight_prep
from .op.weight_prep import weight_prep_cached as _weight_prep_cached
from .op._dispatch import strict_native as _strict_native


def _fused_tails():
    """SR_FUSED_TAILS=0 runs conv / blur / noise+bias+activation as separate autograd nodes (A/B, debugging)."""
    import os

    return os.environ.get("SR_FUSED_TAILS", "1") != "0"


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.dim() == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


class PixelNorm(nn.Module):
    def __init__(self, eps=1e-8):
        super().__init__()
        self.eps = abs(eps)

    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input * input, -1, keepdim=True) + self.eps)


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


_CONST_ROWS = {}


def _const_rows(value, b, n, device):
    """[b, n] tensor filled with `value` (an output scale table of the convolution kernels), made once per shape and
    NEVER dropped: captured phases bake its address into their hipGraph (a size-triggered clear() freed
    tables that replays still read).  A few KB per (value, batch, channels) — the set is bounded by the layer shapes of
    the networks in the process."""
    from .op._dispatch import hold_for_capture

    key = (float(value), int(b), int(n), str(device))
    hit = _CONST_ROWS.get(key)
    if hit is None:
        hit = _CONST_ROWS[key] = torch.full((b, n), float(value), dtype=torch.float32, device=device)
    return hold_for_capture(hit)


def _native_f32(input, weight):
    """Device tensors take this library's kernels only as float32: they walk raw float pointers, so a .double() model, an
    fp16 / bf16 or autocast input must never reach them (silent garbage / out-of-bounds reads)."""
    return input.device.type == "cuda" and input.dtype == torch.float32 and weight.dtype == torch.float32


class EqualConv2d(nn.Module):
    """Plain convolution with equalised learning rate (reference layers.py:204-221)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def _geom(self):
        k = self.weight.shape[2]
        return {(3, 1, 1): "c3", (3, 2, 0): "c3s2", (1, 1, 0): "c1", (1, 2, 0): "c1s2"}.get(
            (k, self.stride, self.padding))

    def forward(self, input, with_bias=True, gain=None):
        """gain (device path, bias-free layers): a constant factor on the output, applied in the convolution's store
        (its per-(sample, channel) output scale) — ResBlock folds its 1/sqrt(2) in here instead of a separate pass."""
        geom = self._geom()
        bias = self.bias if with_bias else None
        if _native_f32(input, self.weight) and geom is not None:
            wt, _ = _weight_prep_cached(self, self.weight, self.scale)
            osc = _const_rows(gain, input.shape[0], self.weight.shape[0], input.device) if gain is not None else None
            assert osc is None or bias is None
            return _conv.conv2d(input, wt, None, osc, bias, geom)
        if input.device.type == "cuda" and _convg.supported(input, self.weight):
            # any other kernel extent / stride / padding (reference layers.py:204-221): the direct kernels of
            # csrc/conv_generic.hip, not MIOpen
            out = _convg.conv2d_generic(input, self.weight * self.scale, bias, self.stride, self.padding)
            return out * gain if gain is not None else out
        if input.device.type == "cuda" and _strict_native():
            raise RuntimeError("SR_STRICT_NATIVE: EqualConv2d on a %s device tensor would run on MIOpen" % input.dtype)
        out = F.conv2d(input, self.weight * self.scale, bias=bias, stride=self.stride, padding=self.padding)
        return out * gain if gain is not None else out

    def forward_stride1(self, input, gain=None, addend=None):
        """The 1x1 convolution applied to an input that is already decimated (see ConvLayer.forward).  addend (bias-free
        layers): a tensor of the output's shape added in the convolution's store where the GEMM-shaped kernel takes the
        call (op.conv.conv1x1_add) — ResBlock's sum of its two branches."""
        wt, _ = _weight_prep_cached(self, self.weight, self.scale)
        osc = _const_rows(gain, input.shape[0], self.weight.shape[0], input.device) if gain is not None else None
        assert osc is None or self.bias is None
        if addend is not None and self.bias is None:
            return _conv.conv1x1_add(input.contiguous(), wt, osc, addend.contiguous())
        out = _conv.conv2d(input, wt, None, osc, self.bias, "c1")
        return out if addend is None else out + addend

    def __repr__(self):
        return "%s(%d, %d, %d, stride=%d, padding=%d)" % (
            self.__class__.__name__, self.weight.shape[1], self.weight.shape[0],
            self.weight.shape[2], self.stride, self.padding)


class EqualLinear(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        if self.activation in (None, "fused_lrelu") and _style.linear_supported(input, self.weight):
            # device tensors: one launch (scale, bias and the mapping network's leaky-ReLU fused)
            return _style.equal_linear(input, self.weight, self.bias, self.scale, self.lr_mul,
                                       self.activation == "fused_lrelu")
        if input.device.type == "cuda" and self.activation in (None, "fused_lrelu") and _strict_native():
            raise RuntimeError("SR_STRICT_NATIVE: EqualLinear input %s (stride %s) is outside sr_linear_fwd and would "
                               "run on rocBLAS" % (tuple(input.shape), tuple(input.stride())))
        if self.activation == "fused_lrelu":
            out = F.linear(input, self.weight * self.scale)
            return fused_leaky_relu(out, self.bias * self.lr_mul)
        bias = self.bias * self.lr_mul if self.bias is not None else None
        out = F.linear(input, self.weight * self.scale, bias=bias)
        if self.activation == "relu":
            out = F.relu(out)
        elif self.activation == "lrelu":
            out = F.leaky_relu(out, negative_slope=0.2)
        elif self.activation == "selu":
            out = F.selu(out)
        elif self.activation == "tanh":
            out = torch.tanh(out)
        return out

    def __repr__(self):
        return "%s(%d, %d)" % (self.__class__.__name__, self.weight.shape[1], self.weight.shape[0])


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True,
                 upsample=False, downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1),
                             upsample_factor=factor)
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def __repr__(self):
        return "%s(%d, %d, %d, upsample=%s, downsample=%s)" % (
            self.__class__.__name__, self.in_channel, self.out_channel, self.kernel_size,
            self.upsample, self.downsample)

    # ---- device tensors: shared weights + operand scaling on the MFMA kernels
    def _wprep(self, want_sq):
        """Tap-major scaled weights (+ the demodulation matrix): one launch per call, or once per weight version for
        a module frozen with op.weight_prep.freeze_prepared_weights (inversion, sampling)."""
        return _weight_prep_cached(self, self.weight, self.scale, want_sq)

    def style_of(self, style):
        """Modulation vector [B, Ci]: from the latent row, or precomputed for all layers at once (op.style_bank)."""
        return style.s if isinstance(style, StylePack) else self.modulation(style)

    def _prepared(self, style, s):
        """(wt, wsq, d): tap-major weights, demodulation matrix and factor — from the StylePack when it carries them."""
        if isinstance(style, StylePack) and style.wt is not None:
            return style.wt, style.wsq, style.d
        # one launch: tap-major scaled weights + the demodulation matrix sum_taps (scale*W)^2 [Ci, Co]
        wt, wsq = self._wprep(self.demodulate)
        d = None
        if self.demodulate:
            if _style.demod_supported(s, wsq):
                d = _style.demod_scale(s, wsq, self.eps)                 # [B, Co]
            else:
                if _strict_native():
                    raise RuntimeError("SR_STRICT_NATIVE: demodulation of shape %s x %s is outside sr_demod_fwd and would "
                                       "run on rocBLAS" % (tuple(s.shape), tuple(wsq.shape)))
                d = torch.rsqrt(torch.matmul(s * s, wsq) + self.eps)
        return wt, wsq, d

    def _forward_mfma(self, input, style, skip_blur=False):
        s = self.style_of(style)                                         # [B, Ci]
        if (self.kernel_size == 1 and not self.demodulate and not self.upsample and not self.downsample
                and _smallconv.supported(input, self.out_channel)):
            # ToRGB: <= 4 output channels -> streaming kernels instead of 128-wide MFMA tiles
            return _smallconv.modulated_conv1x1_small(input, self.weight.view(self.out_channel, self.in_channel), s,
                                                      scale=self.scale)
        k = self.kernel_size
        if k not in (1, 3) or ((self.upsample or self.downsample) and k != 3):
            return self._forward_generic(input, s, skip_blur)
        wt, wsq, d = self._prepared(style, s)
        if self.upsample:
            out = _conv.conv2d(input, wt, s, d, None, "t3s2")
            return out if skip_blur else self.blur(out)      # skip_blur: the caller fuses it (StyledConv)
        if self.downsample:
            return _conv.conv2d(self.blur(input), wt, s, d, None, "c3s2")
        if k == 3:
            return _conv.conv2d(input, wt, s, d, None, "c3")
        return _conv.conv2d(input, wt, s, d, None, "c1")

    def _forward_generic(self, input, s, skip_blur=False):
        """Any other kernel_size (reference layers.py:259-323 takes every odd extent, also up- and down-sampling): the
        shared-weight form  y = d[b, o] * conv(s[b, i] * x, scale * W)  with the modulation and demodulation as
        element-wise operand scalings around the direct kernels of csrc/conv_generic.hip — no grouped convolution, no
        MIOpen.  Equal to the reference's per-sample weights: (scale W s) * rsqrt(sum (scale W s)^2 + eps)."""
        w = self.weight[0] * self.scale                                      # [Co, Ci, k, k]
        x = input * s[:, :, None, None]
        d = None
        if self.demodulate:
            wsq = w.pow(2).sum((2, 3)).t().contiguous()                      # [Ci, Co]
            if _style.demod_supported(s, wsq):
                d = _style.demod_scale(s, wsq, self.eps)                     # [B, Co] (sr_demod_fwd)
            else:                                                            # (Co % 4 != 0: element-wise, not rocBLAS)
                d = torch.rsqrt(((s * s)[:, :, None] * wsq[None]).sum(1) + self.eps)
        if self.upsample:
            out = _convg.conv_transpose2d_generic(x, w.transpose(0, 1), stride=2, padding=0)
        elif self.downsample:
            out = _convg.conv2d_generic(self.blur(x), w, None, 2, 0)
        else:
            out = _convg.conv2d_generic(x, w, None, 1, self.padding)
        if d is not None:
            out = out * d[:, :, None, None]
        return self.blur(out) if (self.upsample and not skip_blur) else out

    def forward_noise_bias_act(self, input, style, noise, noise_weight, act_bias, negative_slope, act_scale):
        """conv -> noise -> bias -> LeakyReLU of a non-upsampling 3x3 layer as one autograd node (device tensors,
        Winograd-eligible shapes); None when that path does not apply (the caller runs the separate operators)."""
        if (input.device.type != "cuda" or self.upsample or self.downsample or self.kernel_size != 3
                or not self.demodulate or act_bias is None or not _fused_tails()):
            return None
        # shape-only eligibility first: the 4^2 .. 16^2 layers fail it, and computing the modulation and the
        # weight prep here would repeat those launches in the caller's fallback
        input = input.contiguous()
        if not _conv.conv_nba_shape_ok(input, self.out_channel, noise):
            return None
        s = self.style_of(style)
        if isinstance(style, StylePack) and style.d is not None:
            wt, wsq, d = style.wt, style.wsq, style.d
            if not _conv.conv_nba_supported(input, wt, noise):
                return None
        else:
            wt, wsq = self._wprep(True)
            if not (_conv.conv_nba_supported(input, wt, noise) and _style.demod_supported(s, wsq)):
                return None
            d = _style.demod_scale(s, wsq, self.eps)
        return _conv.conv2d_nba(input, wt, s, d, noise, noise_weight, act_bias, negative_slope, act_scale)

    def forward_up_noise_bias_act(self, input, style, noise, noise_weight, act_bias, negative_slope, act_scale):
        """Upsampling 3x3 layer: transposed conv -> blur -> noise -> bias -> LeakyReLU as one autograd node
        (device tensors); None when that path does not apply."""
        if (input.device.type != "cuda" or not self.upsample or self.kernel_size != 3 or not self.demodulate
                or act_bias is None or not _fused_tails()):
            return None
        s = self.style_of(style)
        pad = self.blur.pad
        oh = 2 * input.shape[2] + 1 + pad[0] + pad[1] - 3
        ow = 2 * input.shape[3] + 1 + pad[0] + pad[1] - 3
        if isinstance(style, StylePack) and style.d is not None:
            wt, wsq, d = style.wt, style.wsq, style.d
            if not _conv.upconv_nba_supported(input, wt, noise, oh, ow):
                return None
        else:
            wt, wsq = self._wprep(True)
            if not (_conv.upconv_nba_supported(input, wt, noise, oh, ow) and _style.demod_supported(s, wsq)):
                return None
            d = _style.demod_scale(s, wsq, self.eps)
        kernel = self.blur.kernel
        return _conv.upconv_nba(input.contiguous(), wt, s, d, kernel, pad, noise, noise_weight, act_bias,
                                negative_slope, act_scale)

    # ---- CPU tensors: the reference's grouped-convolution formulation (reference layers.py:293-323)
    def _forward_grouped(self, input, style):
        batch, in_channel, height, width = input.shape
        k = self.kernel_size
        s = self.style_of(style).view(batch, 1, in_channel, 1, 1)
        weight = self.scale * self.weight * s
        if self.demodulate:
            weight = weight * torch.rsqrt(weight.pow(2).sum([2, 3, 4], keepdim=True) + self.eps)
        if self.upsample:
            wt = weight.transpose(1, 2).reshape(batch * in_channel, self.out_channel, k, k)
            out = F.conv_transpose2d(input.reshape(1, batch * in_channel, height, width), wt,
                                     padding=0, stride=2, groups=batch)
            return self.blur(out.view(batch, self.out_channel, out.shape[2], out.shape[3]))
        wflat = weight.view(batch * self.out_channel, in_channel, k, k)
        if self.downsample:
            x = self.blur(input)
            out = F.conv2d(x.reshape(1, batch * in_channel, x.shape[2], x.shape[3]), wflat, padding=0,
                           stride=2, groups=batch)
        else:
            out = F.conv2d(input.reshape(1, batch * in_channel, height, width), wflat,
                           padding=self.padding, groups=batch)
        return out.view(batch, self.out_channel, out.shape[2], out.shape[3])

    def forward(self, input, style, skip_blur=False):
        if _native_f32(input, self.weight):
            return self._forward_mfma(input, style, skip_blur)
        if input.device.type == "cuda" and _strict_native():
            raise RuntimeError("SR_STRICT_NATIVE: ModulatedConv2d on %s input / %s weight device tensors would run the "
                               "grouped convolution on MIOpen (the HIP kernels are float32 only)"
                               % (input.dtype, self.weight.dtype))
        # CPU tensors, and device tensors of any other dtype (a .double() model, autocast): the reference's formulation
        return self._forward_grouped(input, style)


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class _BiasSums(torch.autograd.Function):
    """conv.bias + activation.bias of MANY layers in one multi-tensor launch (the layers merge the two so that the bias is
    added once, in the convolution's store or the activation kernel; per layer that was one ~5 us add per forward)."""

    @staticmethod
    def forward(ctx, *biases):
        n = len(biases) // 2
        return tuple(torch._foreach_add(list(biases[:n]), list(biases[n:])))

    @staticmethod
    def backward(ctx, *grads):
        # each merged bias feeds TWO parameters.  Handing the same tensor to both makes AccumulateGrad clone it for one of
        # them — one ~5 us copy launch per layer; one multi-tensor launch makes the second set of tensors for all layers
        live = [g for g in grads if g is not None]
        if live and live[0].is_cuda and not torch.is_grad_enabled():
            copies = iter(torch._foreach_mul(live, 1.0))
            second = tuple(next(copies) if g is not None else None for g in grads)
            return tuple(grads) + second
        return tuple(grads) + tuple(grads)


class BiasBank:
    """`with BiasBank(net):` around a network's forward on device tensors: every ConvLayer with a convolution bias AND an
    activation bias finds its merged bias prepared (one launch for all of them).  SR_BIAS_BANK=0 disables."""

    def __init__(self, net):
        self.net = net
        self.filled = []

    def __enter__(self):
        if os.environ.get("SR_BIAS_BANK", "1") == "0":
            return self
        # the plan is keyed on the identity of the network's ConvLayers: a layer replaced, added or removed after the first
        # forward rebuilds it (a stale plan silently dropped new layers back to the per-layer add)
        layers = [m for m in self.net.modules() if isinstance(m, ConvLayer)]
        key = tuple(id(m) for m in layers)
        cached = getattr(self.net, "_bias_plan", None)
        if cached is None or cached[0] != key:
            cached = (key, [m for m in layers if m.merged_bias_pair() is not None])
            object.__setattr__(self.net, "_bias_plan", cached)
        plan = cached[1]
        pairs = [m.merged_bias_pair() for m in plan]
        live = [(m, p) for m, p in zip(plan, pairs) if p is not None and p[0].device.type == "cuda"
                and p[0].dtype == torch.float32]
        if len(live) < 2:
            return self
        sums = _BiasSums.apply(*([p[0] for _, p in live] + [p[1] for _, p in live]))
        for (m, _), t in zip(live, sums):
            m._bias_sum = t
            self.filled.append(m)
        return self

    def __exit__(self, *exc):
        for m in self.filled:
            m._bias_sum = None
        self.filled = []
        return False


class ConvLayer(nn.Sequential):
    """[Blur] -> EqualConv2d -> [FusedLeakyReLU | ScaledLeakyReLU] (reference layers.py:341-378).
    `activate` is 'lrelu' or anything else for "no activation"; the reference passes False from
    ResBlock.skip, which its own code cannot digest — accepted here.
    Same sub-modules and state_dict keys as the reference; on device tensors `forward` runs two algebraically
    identical shortcuts instead of the module chain:
      * 1x1 stride-2 convolution after the blur (ResBlock.skip): a 1x1 stride-2 convolution only reads every second
        blurred pixel, so the blur is evaluated AT those pixels (upfirdn2d with down = 2: a quarter of the FIR
        work, and its gradient is one zero-insert + FIR pass instead of zero-fill, strided copy and a full blur);
      * the convolution's own bias and the FusedLeakyReLU bias (the reference has both) are added as one vector
        in the activation kernel: their gradients are the same sum, computed once in the activation backward
        instead of a second full-tensor reduction;
      * stride-1 3x3 layers on Winograd-eligible maps apply that bias + LeakyReLU in the convolution's store
        (op.conv.ConvNBAFn, the node StyledConv uses, without noise or modulation)."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False,
                 blur_kernel=[1, 3, 3, 1], bias=True, activate="lrelu"):
        layers = []
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, padding = 2, 0
        else:
            stride, padding = 1, kernel_size // 2
        self.padding = padding
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=padding,
                                  stride=stride, bias=bias))
        if activate == "lrelu":
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)

    def merged_bias_pair(self):
        """(conv.bias, activation.bias) when forward() merges them (device path), else None."""
        mods = [m for m in self if not isinstance(m, Blur)]
        if (len(mods) >= 2 and isinstance(mods[0], EqualConv2d) and isinstance(mods[1], FusedLeakyReLU)
                and mods[0].bias is not None and mods[0]._geom() is not None):
            return mods[0].bias, mods[1].bias
        return None

    def forward(self, input, gain=None):
        """gain (device path only): a constant factor on the layer's output, folded into the activation's gain
        (sqrt(2) * gain) or, without activation and bias, into the convolution's store."""
        if input.device.type != "cuda" or input.dtype != torch.float32:
            out = super().forward(input)
            return out * gain if gain is not None else out
        mods = list(self)
        x = input
        g = 1.0 if gain is None else float(gain)
        if isinstance(mods[0], Blur):
            blur, conv = mods[0], mods[1]
            if conv.weight.shape[2] == 1 and conv.stride == 2 and conv.padding == 0:
                x = upfirdn2d(x, blur.kernel, down=2, pad=blur.pad)        # blur evaluated at the kept pixels only
                if len(mods) == 2 and conv.bias is None and gain is not None:
                    return conv.forward_stride1(x, gain=g)
                x = conv.forward_stride1(x)
                mods = mods[2:]
            else:
                x = blur(x)
                mods = mods[1:]
        if mods and isinstance(mods[0], EqualConv2d):
            conv = mods[0]
            act = mods[1] if len(mods) > 1 and isinstance(mods[1], FusedLeakyReLU) else None
            if act is not None and conv.bias is not None and conv._geom() is not None:
                bias = getattr(self, "_bias_sum", None)            # prepared for the whole network (BiasBank)
                if bias is None:
                    bias = conv.bias + act.bias
                if conv._geom() == "c3" and _fused_tails():
                    # stride-1 3x3 layers of 32^2 .. 256^2 maps: bias + LeakyReLU in the Winograd kernel's store
                    # (the pre-activation tensor is never written or kept for backward)
                    xc = x.contiguous()
                    wt, _ = _weight_prep_cached(conv, conv.weight, conv.scale)
                    if _conv.conv_nba_supported(xc, wt, None):
                        return _conv.conv2d_nba(xc, wt, None, None, None, None, bias, act.negative_slope, act.scale * g)
                x = conv(x, with_bias=False)
                return fused_leaky_relu(x, bias, act.negative_slope, act.scale * g)
            if len(mods) == 1 and conv.bias is None and gain is not None and conv._geom() is not None:
                return conv(x, gain=g)
            x = conv(x)
            mods = mods[1:]
        for m in mods:
            x = m(x)
        return x * g if gain is not None else x


RSQRT2 = 1.0 / math.sqrt(2)


def _resblock_fold():
    return os.environ.get("SR_RESBLOCK_FOLD", "1") != "0"


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1], downsample=True):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=downsample)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=downsample, activate=False,
                              bias=False)

    def forward(self, input):
        mods = list(self.skip)
        if (input.device.type == "cuda" and input.dtype == torch.float32 and len(mods) == 2 and isinstance(mods[0], Blur)
                and isinstance(mods[1], EqualConv2d) and mods[1].weight.shape[2] == 1 and mods[1].stride == 2
                and mods[1].padding == 0 and input.shape[2] == input.shape[3] and tuple(mods[0].kernel.shape) == (4, 4)
                and os.environ.get("SR_SKIP_FUSED", "1") != "0"):
            # the input feeds conv1 and the (blurred, decimated) skip branch: one node whose backward adds the two
            # input gradients inside the up-sampling FIR kernel of the skip branch (op.upfirdn2d.SkipDown)
            same, down = _skip_down(input, mods[0].kernel, mods[0].pad)
            if mods[1].bias is None and _resblock_fold():
                # (a + b) / sqrt(2) = a / sqrt(2) + b / sqrt(2): the factor rides in conv2's activation gain
                # (sqrt(2) * 1/sqrt(2)) and in the skip convolution's store — one full-size pass (the add) instead
                # of two, forward and backward
                # ... and the sum itself in the skip convolution's store
                return mods[1].forward_stride1(down, gain=RSQRT2, addend=self.conv2(self.conv1(same), gain=RSQRT2))
            out = self.conv2(self.conv1(same))
            return (out + mods[1].forward_stride1(down)) / math.sqrt(2)
        if input.device.type == "cuda" and input.dtype == torch.float32 and _resblock_fold():
            return self.conv2(self.conv1(input), gain=RSQRT2) + self.skip(input, gain=RSQRT2)
        out = self.conv2(self.conv1(input))
        return (out + self.skip(input)) / math.sqrt(2)
