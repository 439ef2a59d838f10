"""LPIPS perceptual distance (VGG16, version 0.1) on the MI355X-native operators (SURVEY.md N3 / BASELINE config[4]).

Mirrors the reference's lpips/networks_basic.py:27-112 (`PNetLin(pnet_type='vgg')`, `ScalingLayer`, `NetLinLayer`)
on the trunk slicing of lpips/pretrained_networks.py:97-135, with the SAME module / state_dict names, so the
reference's own files load unchanged:

    scaling_layer            (x - shift) / scale                                   networks_basic.py:94-101
    net.slice1 .. slice5     torchvision vgg16().features[0:4], [4:9], [9:16], [16:23], [23:30]: 13 x
                             [conv3x3 + bias + ReLU], a 2x2 max-pool in front of slices 2-5; taps relu1_2, relu2_2,
                             relu3_3, relu4_3, relu5_3 (64, 128, 256, 512, 512 channels); keys `net.slice1.0.weight` ...
    normalize_tensor         x / (sqrt(sum_c x^2) + 1e-10)                          lpips/__init__.py:42-44
    lin0 .. lin4             Dropout (identity in eval) + 1x1 convolution, no bias, one output channel; keys
                             `lin0.model.1.weight` ... as in lpips/weights/v0.1/vgg.pth   networks_basic.py:103-112
    distance                 sum_k spatial_mean( lin_k( (f0_k - f1_k)^2 ) )          networks_basic.py:62-85

On device tensors every 3x3 convolution runs on the MFMA kernels (op.conv.conv2d, Winograd where eligible),
bias + ReLU is the fused activation kernel (negative_slope = 0, scale = 1), and — against a fixed target, the case of
the inversion loop — normalisation, squared difference, `lin` and the spatial mean of a layer are one fused launch
forward and one backward (op.lpips_layer, csrc/lpips.hip: ~25 ATen launches per layer otherwise, at batch 1 where
every launch is 5 us of a 9 ms step); CPU tensors use torch's own ops.

WEIGHTS.  The five learned heads ARE the reference's (`lpips/weights/v0.1/vgg.pth`, 1 472 floats, BSD-2 LICENSE-LPIPS;
shipped as data in `lpips_heads_v0_1.npz`, written by oracle/make_golden.py) and are loaded by default.  The trunk
is torchvision's ImageNet-pretrained VGG16 in the reference; neither torchvision nor a network exists here, so the
default trunk is a deterministic He-style fill (`synthetic_trunk_state`): the loss has LPIPS's architecture, heads
and cost, not its ImageNet calibration.  `load_trunk_state_dict` takes `torchvision.models.vgg16().features
.state_dict()` when a deployment has it.  tests/golden/lpips_vgg.npz pins this module to the reference's classes
(same trunk fill + real heads): distance, per-layer terms and the input gradient.
"""
import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import synth
from .op import conv as _conv
from .op import fused_leaky_relu
from .op import lpips_layer as _lpips_layer
from .op.weight_prep import weight_prep as _weight_prep

VGG_CFG = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))
# index of each conv inside torchvision's vgg16().features, slice by slice
VGG_FEATURE_INDEX = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))
HEADS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lpips_heads_v0_1.npz")


def synthetic_trunk_state():
    """torchvision-keyed ('0.weight', '0.bias', '2.weight', ...) deterministic stand-in for the pretrained trunk."""
    state, cin, key = {}, 3, 7000
    for widths, idxs in zip(VGG_CFG, VGG_FEATURE_INDEX):
        for cout, i in zip(widths, idxs):
            w = synth.det_normal((cout, cin, 3, 3), key) * np.float32(np.sqrt(2.0 / (cin * 9)))
            state["%d.weight" % i] = torch.from_numpy(w)
            state["%d.bias" % i] = torch.from_numpy(synth.det_normal((cout,), key + 1) * np.float32(0.05))
            cin, key = cout, key + 2
    return state


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


def _fused_trunk():
    return os.environ.get("SR_LPIPS_FUSED_TRUNK", "1") != "0"


class _Conv3x3ReLU(nn.Module):
    """features[i] (Conv2d 3x3 pad 1) fused with features[i+1] (ReLU)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, 3, 3), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        self._prepared = None

    def forward(self, x):
        if x.device.type == "cuda" and x.dtype == torch.float32:
            # frozen weights: the tap-major copy is prepared once per (device, version)
            tag = (self.weight.device, self.weight._version, self.weight.data_ptr())
            if self._prepared is None or self._prepared[0] != tag:
                with torch.no_grad():
                    wt = _weight_prep(self.weight, 1.0)[0]
                    wt._sr_frozen = True                # trunk weights never train: ConvFn caches their adjoints
                    self._prepared = (tag, wt)
            wt = self._prepared[1]
            xc = x.contiguous()
            if _fused_trunk() and _conv.conv_nba_supported(xc, wt, None):
                # Winograd-eligible shapes (9 of the 13 layers at 256^2): bias + ReLU ride in the convolution's store
                return _conv.conv2d_nba(xc, wt, None, None, None, None, self.bias, 0.0, 1.0)
            out = _conv.conv2d(x, wt, None, None, None, "c3")
            return fused_leaky_relu(out, self.bias, 0.0, 1.0)                     # bias + ReLU in one pass
        return F.relu(F.conv2d(x, self.weight, self.bias, padding=1))


class _Slice(nn.ModuleDict):
    """One `sliceN` of pretrained_networks.vgg16: children keyed by their torchvision feature index."""

    def __init__(self, pool, convs):
        super().__init__(convs)
        self.pool = pool

    def forward(self, x):
        if self.pool:
            x = _lpips_layer.max_pool2(x)            # device tensors: sr_maxpool2_* (no MIOpen / ATen pooling kernel)
        for layer in self.values():
            x = layer(x)
        return x


class VGG16Trunk(nn.Module):
    """The five feature slices of torchvision's VGG16 (reference lpips/pretrained_networks.py:97-135)."""

    def __init__(self):
        super().__init__()
        cin = 3
        for s, (widths, idxs) in enumerate(zip(VGG_CFG, VGG_FEATURE_INDEX)):
            convs = {}
            for cout, i in zip(widths, idxs):
                convs[str(i)] = _Conv3x3ReLU(cin, cout)
                cin = cout
            setattr(self, "slice%d" % (s + 1), _Slice(s > 0, convs))
        self.N_slices = 5
        self.load_trunk_state_dict(synthetic_trunk_state())

    @property
    def slices(self):
        return [list(getattr(self, "slice%d" % (s + 1)).values()) for s in range(self.N_slices)]

    def forward(self, x):
        feats = []
        for s in range(self.N_slices):
            x = getattr(self, "slice%d" % (s + 1))(x)
            feats.append(x)
        return feats

    def load_trunk_state_dict(self, features_state):
        """The ImageNet trunk the reference takes from torchvision (pretrained_networks.py:97-135:
        `tv.vgg16(pretrained=True).features`, sliced at Sequential indices 0-3 / 4-8 / 9-15 / 16-22 / 23-29).  Accepts
        either `vgg16().features.state_dict()` ('0.weight', '0.bias', '2.weight', ...) or the whole model's
        `vgg16().state_dict()` ('features.0.weight', ...; `classifier.*` entries are ignored)."""
        if any(k.startswith("features.") for k in features_state):
            features_state = {k[len("features."):]: v for k, v in features_state.items() if k.startswith("features.")}
        want = ["%d.%s" % (i, kind) for idxs in VGG_FEATURE_INDEX for i in idxs for kind in ("weight", "bias")]
        missing = [k for k in want if k not in features_state]
        if missing:
            raise KeyError("VGG16 trunk state is missing %s" % missing[:4])
        with torch.no_grad():
            for layers, idxs in zip(self.slices, VGG_FEATURE_INDEX):
                for layer, i in zip(layers, idxs):
                    layer.weight.copy_(features_state["%d.weight" % i])
                    layer.bias.copy_(features_state["%d.bias" % i])
                    layer._prepared = None


class NetLinLayer(nn.Module):
    """networks_basic.py:103-112: `model` = [Dropout, Conv2d(chn_in, 1, 1, bias=False)] — key `model.1.weight`."""

    def __init__(self, chn_in, chn_out=1, use_dropout=True):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)
        self.model[-1].weight.requires_grad_(False)

    @property
    def weight(self):
        return self.model[-1].weight


class PNetLin(nn.Module):
    """d(in0, in1) -> [B, 1, 1, 1]; inputs in [-1, 1], like the reference's (version 0.1, lpips=True, eval)."""

    def __init__(self, use_dropout=True, heads="default"):
        super().__init__()
        self.chns = [w[-1] for w in VGG_CFG]
        self.L = len(self.chns)
        self.scaling_layer = ScalingLayer()
        self.net = VGG16Trunk()
        for k, c in enumerate(self.chns):
            setattr(self, "lin%d" % k, NetLinLayer(c, use_dropout=use_dropout))
        if heads == "default":
            if not os.path.isfile(HEADS_FILE):
                raise FileNotFoundError("LPIPS heads %s missing (written by oracle/make_golden.py lpips)" % HEADS_FILE)
            with np.load(HEADS_FILE) as z:
                self.load_lin_state_dict({"lin%d.model.1.weight" % k: torch.from_numpy(z["lin%d" % k])
                                          for k in range(self.L)})
        elif heads is not None:
            self.load_lin_state_dict(heads)
        self.eval()                                     # a metric: dropout is never active

    @property
    def lins(self):
        return [getattr(self, "lin%d" % k).weight for k in range(self.L)]

    def features(self, x):
        return [normalize_tensor(f) for f in self.net(self.scaling_layer(x))]

    def per_layer(self, feats0, feats1):
        """[spatial_average(lin_k((f0_k - f1_k)^2))] (networks_basic.py:66-76, spatial=False)."""
        lins = self.lins
        return [spatial_average((((feats0[k] - feats1[k]) ** 2) * lins[k]).sum(1, keepdim=True))
                for k in range(self.L)]

    def distance_to(self, feats1, in0):
        """Distance of `in0` to precomputed `features(in1)` (the target of an optimisation is fixed)."""
        raw = self.net(self.scaling_layer(in0))
        lins = self.lins
        if all(_lpips_layer.supported(raw[k], feats1[k], lins[k]) for k in range(self.L)):
            # device tensors, fixed target and heads: normalisation, squared difference, `lin` and the spatial mean
            # of a layer are one launch forward and one backward (op/lpips_layer.py) instead of ~25
            res = [_lpips_layer.lpips_layer(raw[k], feats1[k], lins[k]) for k in range(self.L)]
        else:
            res = self.per_layer([normalize_tensor(f) for f in raw], feats1)
        val = res[0]
        for r in res[1:]:
            val = val + r
        return val

    def forward(self, in0, in1, retPerLayer=False):
        feats1 = self.features(in1)
        if retPerLayer:
            res = self.per_layer(self.features(in0), feats1)
            return sum(res[1:], res[0]), res
        return self.distance_to(feats1, in0)

    def load_lin_state_dict(self, state):
        """LPIPS weights/v0.1/vgg.pth: keys 'lin0.model.1.weight' ... 'lin4.model.1.weight' [1, C, 1, 1]."""
        with torch.no_grad():
            for k in range(self.L):
                getattr(self, "lin%d" % k).weight.copy_(state["lin%d.model.1.weight" % k])
