"""LPIPS-shaped perceptual distance on the MI355X-native operators (SURVEY.md N3 / BASELINE config[4]).

Structure of the reference's lpips/networks_basic.py:27-92 (`PNetLin`, pnet_type='vgg', version 0.1) with the
trunk of lpips/pretrained_networks.py:97-135:

    ScalingLayer      (x - shift) / scale                                   networks_basic.py:94-101
    vgg16 trunk       13 x [conv3x3 + bias + ReLU], max-pool before slices 2-5; taps relu1_2, relu2_2,
                      relu3_3, relu4_3, relu5_3 (64, 128, 256, 512, 512 channels)
    normalize_tensor  x / (sqrt(sum_c x^2) + 1e-10)                          lpips/__init__.py
    lin_k             1x1 convolution, no bias, one output channel           networks_basic.py:103-112
    distance          sum_k spatial_mean( lin_k( (f0_k - f1_k)^2 ) )

On device tensors every 3x3 convolution runs on the MFMA kernels (op.conv.conv2d, Winograd where eligible) and
bias + ReLU is the fused activation kernel (negative_slope = 0, scale = 1); CPU tensors use torch's own ops.

WEIGHTS: the reference takes the trunk from torchvision's pretrained VGG16 and the five linear heads from
lpips/weights/v0.1/vgg.pth.  Neither torchvision nor a network is available here, so the default construction
fills both with a deterministic He-style initialisation (non-negative heads, like the learned ones): the loss has
LPIPS's architecture and cost, not its calibration.  `load_trunk_state_dict` / `load_lin_state_dict` accept the
real tensors (torchvision `features.N.*` keys, LPIPS `linK.model.1.weight` keys) when a deployment has them.
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import synth
from .op import conv as _conv
from .op import fused_leaky_relu
from .op.weight_prep import weight_prep as _weight_prep

VGG_CFG = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))
# index of each conv inside torchvision's vgg16().features, slice by slice
VGG_FEATURE_INDEX = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, inp):
        return (inp - self.shift) / self.scale


def normalize_tensor(x, eps=1e-10):
    return x / (torch.sqrt(torch.sum(x * x, dim=1, keepdim=True)) + eps)


def spatial_average(x, keepdim=True):
    return x.mean([2, 3], keepdim=keepdim)


class _Conv3x3ReLU(nn.Module):
    def __init__(self, cin, cout, key):
        super().__init__()
        w = synth.det_normal((cout, cin, 3, 3), key) * np.float32(np.sqrt(2.0 / (cin * 9)))
        self.weight = nn.Parameter(torch.from_numpy(w), requires_grad=False)
        self.bias = nn.Parameter(torch.from_numpy(synth.det_normal((cout,), key + 1) * np.float32(0.05)),
                                 requires_grad=False)
        self._prepared = None

    def forward(self, x):
        if x.device.type == "cuda" and x.dtype == torch.float32:
            # frozen weights: the tap-major copy is prepared once per (device, version)
            tag = (self.weight.device, self.weight._version, self.weight.data_ptr())
            if self._prepared is None or self._prepared[0] != tag:
                with torch.no_grad():
                    self._prepared = (tag, _weight_prep(self.weight, 1.0)[0])
            out = _conv.conv2d(x, self._prepared[1], None, None, None, "c3")
            return fused_leaky_relu(out, self.bias, 0.0, 1.0)                     # bias + ReLU in one pass
        return F.relu(F.conv2d(x, self.weight, self.bias, padding=1))


class VGG16Trunk(nn.Module):
    """The five feature slices of torchvision's VGG16 (reference lpips/pretrained_networks.py:97-135)."""

    def __init__(self):
        super().__init__()
        self.slices = nn.ModuleList()
        cin, key = 3, 7000
        for widths in VGG_CFG:
            layers = []
            for cout in widths:
                layers.append(_Conv3x3ReLU(cin, cout, key))
                cin, key = cout, key + 2
            self.slices.append(nn.ModuleList(layers))

    def forward(self, x):
        feats = []
        for i, layers in enumerate(self.slices):
            if i > 0:
                x = F.max_pool2d(x, 2, 2)
            for layer in layers:
                x = layer(x)
            feats.append(x)
        return feats

    def load_trunk_state_dict(self, features_state):
        """torchvision vgg16().features.state_dict() ('0.weight', '0.bias', '2.weight', ...)."""
        with torch.no_grad():
            for layers, idxs in zip(self.slices, VGG_FEATURE_INDEX):
                for layer, i in zip(layers, idxs):
                    layer.weight.copy_(features_state["%d.weight" % i])
                    layer.bias.copy_(features_state["%d.bias" % i])
                    layer._prepared = None


class PNetLin(nn.Module):
    """d(in0, in1) -> [B, 1, 1, 1]; inputs in [-1, 1], like the reference's (version 0.1, lpips=True)."""

    def __init__(self):
        super().__init__()
        self.chns = [w[-1] for w in VGG_CFG]
        self.L = len(self.chns)
        self.scaling_layer = ScalingLayer()
        self.net = VGG16Trunk()
        self.lins = nn.ParameterList()
        for k, c in enumerate(self.chns):
            w = np.abs(synth.det_normal((1, c, 1, 1), 7100 + k)) / np.float32(c)
            self.lins.append(nn.Parameter(torch.from_numpy(w.astype(np.float32)), requires_grad=False))

    def features(self, x):
        return [normalize_tensor(f) for f in self.net(self.scaling_layer(x))]

    def distance_to(self, feats1, in0):
        """Distance of `in0` to precomputed `features(in1)` (the target of an optimisation is fixed)."""
        feats0 = self.features(in0)
        val = 0
        for k in range(self.L):
            diff = (feats0[k] - feats1[k]) ** 2
            val = val + spatial_average((diff * self.lins[k]).sum(1, keepdim=True))
        return val

    def forward(self, in0, in1):
        return self.distance_to(self.features(in1), in0)

    def load_lin_state_dict(self, state):
        """LPIPS weights/v0.1/vgg.pth: keys 'lin0.model.1.weight' ... 'lin4.model.1.weight' [1, C, 1, 1]."""
        with torch.no_grad():
            for k in range(self.L):
                self.lins[k].copy_(state["lin%d.model.1.weight" % k])
