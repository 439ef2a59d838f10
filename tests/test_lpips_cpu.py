"""CPU: `lpips.PNetLin` against tests/golden/lpips_vgg.npz — the reference's own PNetLin / ScalingLayer / NetLinLayer /
normalize_tensor (lpips/networks_basic.py:27-112, lpips/__init__.py:42-44, pretrained_networks.py:97-135) run in the
authoring container with the real learned heads lpips/weights/v0.1/vgg.pth (oracle/make_golden.py gold_lpips)."""
import numpy as np
import torch

from stylerenderer_amd import lpips
from util import rel_err

T = torch.from_numpy


def test_default_heads_are_the_reference_heads(golden):
    gold = golden("lpips_vgg")
    net = lpips.PNetLin()
    assert not net.training
    for k, c in enumerate((64, 128, 256, 512, 512)):
        assert net.lins[k].shape == (1, c, 1, 1)
        assert np.array_equal(net.lins[k].numpy(), gold["lin%d" % k])
    assert float(net.lins[0].min()) > 0                                   # learned heads are non-negative
    assert np.array_equal(net.scaling_layer.shift.numpy(), gold["shift"])
    assert np.array_equal(net.scaling_layer.scale.numpy(), gold["scale"])


def test_state_dict_names_follow_the_reference():
    """`lin0.model.1.weight` (lpips/weights/v0.1/vgg.pth) and `net.slice1.0.weight` (pretrained_networks.vgg16)."""
    net = lpips.PNetLin()
    keys = list(net.state_dict())
    assert keys[:2] == ["scaling_layer.shift", "scaling_layer.scale"]
    want = ["net.slice%d.%d.%s" % (s + 1, i, k) for s, idx in enumerate(lpips.VGG_FEATURE_INDEX) for i in idx
            for k in ("weight", "bias")]
    assert keys[2:2 + 26] == want
    assert keys[28:] == ["lin%d.model.1.weight" % k for k in range(5)]
    # the reference's checkpoint loads the way dist_model.py does it
    heads = {"lin%d.model.1.weight" % k: torch.full((1, c, 1, 1), 0.25) for k, c in enumerate(net.chns)}
    res = net.load_state_dict(heads, strict=False)
    assert not res.unexpected_keys and float(net.lin3.model[1].weight.max()) == 0.25


def test_distance_per_layer_and_gradient_match_the_reference(golden):
    gold = golden("lpips_vgg")
    net = lpips.PNetLin()
    in0 = T(gold["in0"]).requires_grad_(True)
    val, res = net(in0, T(gold["in1"]), retPerLayer=True)
    assert val.shape == (2, 1, 1, 1)
    e = rel_err(val.detach().numpy(), gold["value"])
    assert e < 2e-5, e
    per = np.stack([r.detach().numpy().reshape(-1) for r in res], 0)
    # the reference accumulates in place (`val = res[0]; val += res[l]`, networks_basic.py:78-80): its returned res[0]
    # IS the total; layers 1-4 are the per-layer terms
    e = rel_err(per[1:], gold["per_layer"][1:])
    assert e < 2e-5, e
    assert np.array_equal(gold["per_layer"][0], gold["value"].reshape(-1))
    assert rel_err(per.sum(0), gold["value"].reshape(-1)) < 2e-5
    (g,) = torch.autograd.grad(val.sum(), in0)
    e = float(np.linalg.norm(g.numpy() - gold["grad_in0"]) / np.linalg.norm(gold["grad_in0"]))
    assert e < 1e-4, e
    feats = net.features(T(gold["in1"]))
    got = np.array([float(f.abs().mean()) for f in feats])
    assert np.allclose(got, gold["feat_absmean"], rtol=1e-4)
    # distance_to(features(target), x) == forward(x, target)
    with torch.no_grad():
        d2 = net.distance_to(feats, T(gold["in0"]))
    assert torch.allclose(d2, val.detach(), rtol=1e-6)


def test_trunk_loads_from_torchvision_keyed_state_and_reproduces_the_reference(golden):
    """`load_trunk_state_dict` executed (never run): a trunk that was overwritten is restored from a
    state dict with torchvision's OWN key names (`features.N.weight`, plus the `classifier.*` entries a whole-model
    checkpoint carries, reference lpips/pretrained_networks.py:97-135) and then reproduces the reference's distance."""
    gold = golden("lpips_vgg")
    net = lpips.PNetLin()
    with torch.no_grad():
        for p in net.net.parameters():
            p.normal_(0, 0.02)                                  # not the trunk the fixture was made with
        scrambled = net(T(gold["in0"]), T(gold["in1"]))
    assert rel_err(scrambled.numpy(), gold["value"]) > 1e-2
    feat = lpips.synthetic_trunk_state()                        # '0.weight', '0.bias', '2.weight', ...
    whole = {"features." + k: v for k, v in feat.items()}
    whole.update({"classifier.0.weight": torch.zeros(8, 8), "classifier.0.bias": torch.zeros(8)})
    net.net.load_trunk_state_dict(whole)
    with torch.no_grad():
        val = net(T(gold["in0"]), T(gold["in1"]))
    assert rel_err(val.numpy(), gold["value"]) < 2e-5
    net2 = lpips.PNetLin()
    net2.net.load_trunk_state_dict(feat)                            # the `.features.state_dict()` form
    for a, b in zip(net.net.parameters(), net2.net.parameters()):
        assert torch.equal(a, b)
    import pytest

    with pytest.raises(KeyError):
        net.net.load_trunk_state_dict({"features.0.weight": feat["0.weight"]})
