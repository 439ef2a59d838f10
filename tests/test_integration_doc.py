"""INTEGRATION.md §B is executable: the reference-side ctypes binding shown there (the stub a maintainer would put
in place of the three `torch.utils.cpp_extension.load(...)` calls, reference op/fused_act.py:11-17,
op/upfirdn2d.py:10-16, op/rasterize.py:10-16) is extracted from the document, exec'ed, and its three modules are
driven with the arguments the reference's autograd Functions pass (op/fused_bias_act.cpp:5-33, op/upfirdn2d.cpp:24-87,
op/rasterize.cpp:97-241: forward AND backward, fp32 / fp64) against the CPU oracle and the reference's own
known-answer case — so the documented binding cannot drift from
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
    names = sorted(set(names) | {"sr_rasterize_%s_%s" % (k, t) for k in ("forward", "forward_cpu", "backward",
                                                                             "backward_cpu") for t in ("f32", "f64")})
    assert {"sr_fused_bias_act", "sr_upfirdn2d", "sr_rasterize_scratch_bytes"} <= set(names)
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


def _reference_style_rasterize(rasterize_op):
    """The composition the reference's autograd Function makes of the two extension calls (op/rasterize.py:17-80:
    interpolate with index / coeff in the forward; dcoeff, the [1x3]@[3x9] product and two scatters in the
    backward), restated with index_add_ in place of the COO sparse matmul.  It touches the extension ONLY through
    `rasterize_op.forward` / `rasterize_op.backward`, i.e. through the binding block of the document."""

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, v, tex, tri, h):
            ind, coeff = rasterize_op.forward(v.contiguous(), tri.contiguous(), h, 0, False, 1e-6)
            ctx.save_for_backward(v, tex, ind, coeff)
            c = tex.shape[-1]
            picked = tex.reshape(-1, c)[ind.reshape(-1)].view(tuple(ind.shape) + (c,))
            return (picked * coeff.unsqueeze(-1)).sum(-2)

        @staticmethod
        def backward(ctx, go):
            v, tex, ind, coeff = ctx.saved_tensors
            c = tex.shape[-1]
            flat = ind.reshape(-1)
            dcoeff = rasterize_op.backward(v.contiguous(), ind, False, 1e-6)
            picked = tex.reshape(-1, c)[flat].view(tuple(ind.shape) + (c,))
            dl_dw = (go.unsqueeze(-2) * picked).sum(-1)
            per_corner = torch.matmul(dl_dw.unsqueeze(-2), dcoeff).reshape(-1, 3)
            gv = torch.zeros_like(v).view(-1, 3).index_add_(0, flat, per_corner).view_as(v)
            gt = torch.zeros_like(tex).view(-1, c).index_add_(0, flat, (go.unsqueeze(-2) * coeff.unsqueeze(-1))
                                                              .reshape(-1, c)).view_as(tex)
            return gv, gt, None, None

    return Fn.apply


def _run_reference_kat(ns, dev):
    """The reference's only known-answer test (op/rasterize.py:83-107: one float64 triangle on a 5 x 5 grid, values in
    raster_kat.npz written by the reference itself) through the documented binding: ids, barycentric weights and
    d(coeff)/d(vertex) bit for bit, interpolated output and both gradients, then the reference's own gradcheck call."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "raster_kat.npz"), allow_pickle=False)
    v = torch.from_numpy(g["v"]).to(dev)
    f = torch.from_numpy(g["f"]).to(dev)
    tex = torch.from_numpy(g["tex"]).to(dev)
    op = ns["rasterize_op"]
    index, coeff = op.forward(v, f, 5, 0, False, 1e-6)
    assert index.dtype == torch.int64 and coeff.dtype == torch.float64
    assert np.array_equal(index.cpu().numpy(), g["index"].astype(np.int64))
    assert bits_equal(coeff.cpu().numpy(), g["coeff"])
    dcoeff = op.backward(v, index, False, 1e-6)
    assert tuple(dcoeff.shape) == (1, 5, 5, 3, 9) and bits_equal(dcoeff.cpu().numpy(), g["dcoeff"])
    # shared vertices [n, 3] + shared topology [f, 3]: un-batched outputs (op/rasterize.cpp:103-121, 189-202)
    i2, c2 = op.forward(v[0].contiguous(), f, 5, 0, False, 1e-6)
    assert tuple(i2.shape) == (5, 5, 3) and np.array_equal(i2.cpu().numpy(), g["index"][0].astype(np.int64))
    assert bits_equal(c2.cpu().numpy(), g["coeff"][0])
    d2 = op.backward(v[0].contiguous(), i2, False, 1e-6)
    assert tuple(d2.shape) == (5, 5, 3, 9) and bits_equal(d2.cpu().numpy(), g["dcoeff"][0])
    # the autograd composition of the reference over the two calls
    fn = _reference_style_rasterize(op)
    vv, tt = v.clone().requires_grad_(True), tex.clone().requires_grad_(True)
    out = fn(vv, tt, f, 5)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=0, atol=1e-15)
    out.backward(torch.from_numpy(g["grad_out"]).to(dev))
    # the reference scatters through a float32 COO matmul (op/rasterize.py:63-64, 75-76 `.type(torch.float32)`): its
    # stored gradients carry fp32 rounding, this composition stays in float64
    np.testing.assert_allclose(vv.grad.cpu().numpy(), g["grad_v"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(tt.grad.cpu().numpy(), g["grad_tex"], rtol=1e-6, atol=1e-7)
    x = torch.cat((v, tex), -1).clone().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda x: fn(x[:, :, :3], x[:, :, 3:], f, 5), x, eps=1e-6, atol=1e-6)
    # float32 backward against the C oracle
    v32 = v.float()
    i32, _ = op.forward(v32, f, 5, 0, False, 1e-6)
    d32 = op.backward(v32, i32, False, 1e-6)
    want = raster.backward_dcoeff(g["v"].astype(np.float32), i32.cpu().numpy(), False, 1e-6)
    assert bits_equal(d32.cpu().numpy(), want)


def test_binding_reference_kat_forward_backward_cpu_tensors():
    _run_reference_kat(load_binding(), "cpu")


@pytest.mark.gpu
def test_binding_reference_kat_forward_backward_device():
    _run_reference_kat(load_binding(), "cuda")


@pytest.mark.gpu
def test_binding_rasterizer_backward_device_vs_oracle_mesh():
    """rasterize_op.backward of the binding on a real mesh, fp32 and fp64, bit-exact against the C oracle."""
    from stylerenderer_amd import synth

    ns = load_binding()
    v0, tri = synth.uv_ellipsoid(16, 14)
    v = synth.random_poses(v0, 3, seed=4)
    for dt in (np.float32, np.float64):
        vt = torch.from_numpy(v.astype(dt)).cuda()
        index, coeff = ns["rasterize_op"].forward(vt, torch.from_numpy(tri).cuda(), 48, 48, False, 1e-6)
        wi, wc, _ = raster.forward_buffers(v.astype(dt), tri, 48, 48, False, 1e-6)
        assert np.array_equal(index.cpu().numpy(), wi) and bits_equal(coeff.cpu().numpy(), wc)
        d = ns["rasterize_op"].backward(vt, index, False, 1e-6)
        assert bits_equal(d.cpu().numpy(), raster.backward_dcoeff(v.astype(dt), wi, False, 1e-6))


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
