"""INTEGRATION.md §B is executable: the reference-side ctypes binding shown there (the stub a maintainer would put
in place of the three `torch.utils.cpp_extension.load(...)` calls, reference op/fused_act.py:11-17,
op/upfirdn2d.py:10-16, op/rasterize.py:10-16) is extracted from the document, exec'ed, and its three modules are
driven with the arguments the reference's autograd Functions pass (op/fused_bias_act.cpp:5-33, op/upfirdn2d.cpp:24-87,
op/rasterize.cpp:97-178) against the CPU oracle — so the documented binding cannot drift from
include/stylerenderer_amd.h.  The only edit made to the block is the library path (a bare soname needs
LD_LIBRARY_PATH, which cannot be changed after the interpreter started)."""
import os
import re

import numpy as np
import pytest
import torch

import ops_np
import raster
from util import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "stylerenderer_amd", "libstylerenderer_hip.so")


def binding_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    src = [b for b in blocks if "# op/_hip.py" in b]
    assert len(src) == 1, "INTEGRATION.md must hold exactly one `# op/_hip.py` binding block"
    assert 'ctypes.CDLL("libstylerenderer_hip.so")' in src[0]
    return src[0].replace('ctypes.CDLL("libstylerenderer_hip.so")', "ctypes.CDLL(%r)" % LIB)


def load_binding():
    ns = {}
    exec(compile(binding_source(), "INTEGRATION.md#op/_hip.py", "exec"), ns)
    return ns


def test_binding_block_declares_only_exported_symbols_with_header_arity():
    """CPU: every `_L.sr_*` the block touches is exported, and its argtypes count equals the parameter count of the
    declaration in include/stylerenderer_amd.h."""
    src = binding_source()
    header = open(os.path.join(ROOT, "include", "stylerenderer_amd.h")).read()
    ns = load_binding()
    names = sorted(set(re.findall(r"_L\.(sr_\w+)", src)))
    assert {"sr_fused_bias_act", "sr_upfirdn2d", "sr_rasterize_forward_f32", "sr_rasterize_forward_cpu_f32",
            "sr_rasterize_scratch_bytes"} <= set(names)
    for name in names:
        fn = getattr(ns["_L"], name)                       # AttributeError = not exported
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, header, flags=re.S)
        assert m, "%s is not declared in the header" % name
        n_params = len([p for p in m.group(1).split(",") if p.strip()])
        if fn.argtypes is not None:
            assert len(fn.argtypes) == n_params, (name, len(fn.argtypes), n_params)


def test_binding_rasterizer_cpu_tensors_vs_oracle():
    """CPU tensors take the library's host loops (the reference's extension serves them too): bit-exact."""
    from stylerenderer_amd import synth

    ns = load_binding()
    v0, tri = synth.uv_ellipsoid(16, 14)
    v = synth.random_poses(v0, 2, seed=4)
    index, coeff = ns["rasterize_op"].forward(torch.from_numpy(v), torch.from_numpy(tri), 40, 40, False, 1e-6)
    wi, wc, _ = raster.forward_buffers(v, tri, 40, 40, False, 1e-6)
    assert np.array_equal(index.numpy(), wi) and bits_equal(coeff.numpy(), wc)


@pytest.mark.gpu
def test_binding_runs_the_three_operators_bit_exact_vs_oracle():
    from stylerenderer_amd import synth

    ns = load_binding()
    dev = "cuda"
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    # fused bias + leaky ReLU, the four (act, grad) codes of reference op/fused_act.py:20-72
    x, b, ref = synth.det_normal((3, 8, 17, 19), 5), synth.det_normal((8,), 6), synth.det_normal((3, 8, 17, 19), 7)
    empty = torch.empty(0, device=dev)
    for act, grad in ((3, 0), (3, 1), (3, 2), (1, 0)):
        y = ns["fused"].fused_bias_act(T(x), T(b), T(ref) if grad == 1 else empty, act, grad, 0.2, 2 ** 0.5)
        assert bits_equal(y.cpu().numpy(), ops_np.fused_bias_act(x, b, ref if grad == 1 else None, act, grad))
    y = ns["fused"].fused_bias_act(T(x), empty, empty, 3, 0, 0.2, 2 ** 0.5)
    assert bits_equal(y.cpu().numpy(), ops_np.fused_bias_act(x, None, None, 3, 0))
    # upfirdn2d on the [major, h, w, 1] view the reference's Function hands over (op/upfirdn2d.py:100-108)
    k = ops_np.make_blur_kernel((1, 3, 3, 1), 4.0)
    for (up, down, pad) in ((1, 1, (2, 1)), (2, 1, (2, 1)), (1, 2, (1, 1))):
        xin = synth.det_normal((6, 18, 20, 1), 9)
        out = ns["upfirdn2d_op"].upfirdn2d(T(xin), T(k), up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        want = ops_np.upfirdn2d_full(xin[..., 0][:, None], k, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        assert bits_equal(out.cpu().numpy()[..., 0], np.ascontiguousarray(want[:, 0]))
    # rasterizer forward on device tensors: ids and barycentric weights
    v0, tri = synth.uv_ellipsoid(16, 14)
    v = synth.random_poses(v0, 3, seed=4)
    index, coeff = ns["rasterize_op"].forward(T(v), T(tri), 48, 48, False, 1e-6)
    wi, wc, _ = raster.forward_buffers(v, tri, 48, 48, False, 1e-6)
    assert np.array_equal(index.cpu().numpy(), wi) and bits_equal(coeff.cpu().numpy(), wc)
