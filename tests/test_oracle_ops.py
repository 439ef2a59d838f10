"""CPU: numpy restatements (oracle/ops_np.py) against the golden vectors produced by the
reference's Python CPU branch (fused act, upfirdn2d_native, ModulatedConv2d)."""
import numpy as np
import pytest

import ops_np
from util import max_ulp, rel_err

from make_golden_cases import UFD_TAGS


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_fused_act(golden, tag):
    g = golden("fused_act")
    x, b, gy = g[tag + "_x"], g[tag + "_bias"], g[tag + "_gy"]
    y = ops_np.fused_leaky_relu(x, b)
    assert np.array_equal(y, g[tag + "_y"])                   # same three roundings as torch
    gx, gb = ops_np.fused_leaky_relu_backward(gy, y)
    # the CUDA kernel computes (g * alpha) * scale (reference op/fused_bias_act_kernel.cu:35-41); torch
    # autograd on the CPU branch computes (g * scale) * alpha: two roundings in a different order: <= 2 ulp
    assert max_ulp(gx, g[tag + "_gx"]) <= 2
    assert np.allclose(gb, g[tag + "_gb"], rtol=1e-5, atol=1e-6)
    ggo = ops_np.fused_leaky_relu_double_backward(g[tag + "_ggx"], g[tag + "_ggb"], y)
    assert max_ulp(ggo, g[tag + "_ggo"]) <= 2


@pytest.mark.parametrize("tag", UFD_TAGS)
def test_upfirdn2d(golden, tag):
    g = golden("upfirdn2d")
    up, down, p0, p1 = [int(t) for t in g[tag + "_prm"]]
    x, k = g[tag + "_x"], g[tag + "_k"]
    y = ops_np.upfirdn2d(x, k, up, down, (p0, p1))
    assert y.shape == g[tag + "_y"].shape
    assert np.abs(y - g[tag + "_y"]).max() <= 4e-7 * max(1.0, np.abs(g[tag + "_y"]).max())
    gx = ops_np.upfirdn2d_backward(g[tag + "_gy"], k, x.shape, up, down, (p0, p1))
    assert gx.shape == x.shape
    assert np.abs(gx - g[tag + "_gx"]).max() <= 4e-7 * max(1.0, np.abs(g[tag + "_gx"]).max())


@pytest.mark.parametrize("tag", ["plain", "up", "rgb"])
def test_modulated_conv(golden, tag):
    import torch

    from stylerenderer_amd import synth

    g = golden("modconv")
    cfg = {"plain": (8, 6, 3), "up": (8, 6, 3), "rgb": (8, 3, 1)}[tag]
    ci, co, k = cfg
    # rebuild the parameters the fixture was generated with (pure function of key names)
    sd = {"weight": torch.empty(1, co, ci, k, k), "modulation.weight": torch.empty(ci, 16),
          "modulation.bias": torch.empty(ci)}
    synth.fill_state_dict(sd, salt=31)
    w, mw, mb = (sd[n].numpy() for n in ("weight", "modulation.weight", "modulation.bias"))
    style = g[tag + "_s"].astype(np.float64) @ (mw.astype(np.float64).T / np.sqrt(16)) + mb
    y = ops_np.modulated_conv2d(g[tag + "_x"], w, style, demodulate=(tag != "rgb"),
                                upsample=(tag == "up"),
                                blur_kernel=ops_np.make_blur_kernel((1, 3, 3, 1), 4.0))
    assert y.shape == g[tag + "_y"].shape
    assert rel_err(y, g[tag + "_y"]) < 5e-6


@pytest.mark.parametrize("tag,demod,up", [("m5", True, False), ("m5up", True, True), ("m5nodemod", False, False)])
def test_modulated_conv_kernel_size_5(golden, tag, demod, up):
    """The numpy restatement at a kernel size the matrix-core kernels do not take (reference layers.py:259-323 with
    kernel_size 5: padding 2; up-sampling blur pad (0, 0)) against the reference's layer (conv_generic.npz)."""
    import torch

    from stylerenderer_amd import synth

    g = golden("conv_generic")
    ci, co, k = 8, 12, 5
    sd = {"weight": torch.empty(1, co, ci, k, k), "modulation.weight": torch.empty(ci, 16),
          "modulation.bias": torch.empty(ci)}
    synth.fill_state_dict(sd, salt=35)
    w, mw, mb = (sd[n].numpy() for n in ("weight", "modulation.weight", "modulation.bias"))
    x, s = synth.det_normal((2, 8, 10, 10), 36), synth.det_normal((2, 16), 37)
    style = s.astype(np.float64) @ (mw.astype(np.float64).T / np.sqrt(16)) + mb
    y = ops_np.modulated_conv2d(x, w, style, demodulate=demod, upsample=up,
                                blur_kernel=ops_np.make_blur_kernel((1, 3, 3, 1), 4.0), blur_pad=(0, 0))
    assert y.shape == g[tag + "_y"].shape
    assert rel_err(y, g[tag + "_y"]) < 5e-6
