"""GPU: MFMA implicit-GEMM convolution against a float64 torch-CPU convolution (the third-party
arithmetic the reference itself relies on).  Tolerance: the MFMA is an exact fp32
fma chain, so the error is fp32 round-off of a K-term dot product: |err| <= 2e-6 * sum|a*b| here."""
import numpy as np
import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_conv(x, w_oihw, iscale, oscale, obias, stride, pad, transposed):
    """float64 CPU reference. w_oihw: [N, C, k, k] correlation weights (or, transposed, the
    [C, N, k, k] layout conv_transpose2d takes)."""
    x = x.double()
    if iscale is not None:
        x = x * iscale.double()[:, :, None, None]
    if transposed:
        y = F.conv_transpose2d(x, w_oihw.double(), stride=stride, padding=pad)
    else:
        y = F.conv2d(x, w_oihw.double(), stride=stride, padding=pad)
    if oscale is not None:
        y = y * oscale.double()[:, :, None, None]
    if obias is not None:
        y = y + obias.double()[None, :, None, None]
    return y


def to_taps(w, transposed):
    """[N,C,k,k] (or [C,N,k,k] when transposed) -> [k*k, C, N]."""
    if transposed:
        c, n, k, _ = w.shape
        return w.permute(2, 3, 0, 1).reshape(k * k, c, n).contiguous()
    n, c, k, _ = w.shape
    return w.permute(2, 3, 1, 0).reshape(k * k, c, n).contiguous()


CASES = [
    # B, C, N, H, W, k, stride, pad, transposed
    (2, 8, 6, 8, 8, 3, 1, 1, False),
    (3, 20, 130, 4, 4, 3, 1, 1, False),
    (16, 16, 8, 4, 4, 3, 1, 1, False),
    (2, 5, 7, 16, 16, 3, 1, 1, False),
    (1, 9, 4, 37, 41, 3, 1, 1, False),
    (2, 12, 140, 64, 64, 3, 1, 1, False),
    (4, 3, 3, 64, 64, 3, 1, 1, False),           # map-head layers: streaming weight gradient (k_wgrad_small3)
    (2, 3, 4, 33, 47, 3, 1, 1, False),
    (5, 4, 2, 128, 128, 3, 1, 1, False),         # more pixels than one pass of its 1024 blocks
    (1, 1, 1, 5, 3, 3, 1, 1, False),
    (2, 8, 6, 9, 9, 3, 2, 0, False),
    (2, 10, 5, 33, 33, 3, 2, 0, False),
    (1, 7, 3, 65, 65, 3, 2, 0, False),
    (2, 8, 6, 65, 65, 3, 2, 0, False),           # rotating-lead 16-byte halo DMA, 32 x 4 patches
    (1, 12, 130, 33, 33, 3, 2, 0, False),        # same, 16 x 8 patches, two channel tiles, channel tail (12 = 3 x 4)
    (2, 16, 64, 129, 129, 3, 2, 0, False),       # same, several tiles per row, split over more chunks
    (2, 8, 6, 4, 4, 3, 2, 0, True),
    (2, 8, 6, 8, 8, 3, 2, 0, True),
    (1, 11, 9, 16, 16, 3, 2, 0, True),
    (1, 6, 130, 32, 32, 3, 2, 0, True),
    (2, 8, 130, 32, 32, 3, 2, 0, True),          # fused four-phase kernel (interior) + border strips
    (1, 12, 64, 8, 64, 3, 2, 0, True),
    (2, 128, 32, 16, 16, 3, 2, 0, True),         # weight gradient: DMA-staged stride-2 kernel (k_wgrad_s2_dma)
    (3, 256, 64, 32, 32, 3, 2, 0, True),         # same, two channel tiles each way, several patches per slice
    (2, 32, 128, 33, 33, 3, 2, 0, False),        # same kernel, down-sampling convolution
    (2, 8, 3, 8, 8, 1, 1, 0, False),
    (2, 16, 5, 32, 32, 1, 1, 0, False),
    (4, 3, 128, 32, 32, 1, 1, 0, False),         # from-RGB layer: weight gradient on the streaming row-product kernel
    (2, 3, 64, 16, 8, 1, 1, 0, False),
    (2, 6, 4, 9, 9, 1, 2, 0, False),
]


def test_stride2_rotating_lead_dma_equals_dword_dma(monkeypatch):
    """The 16-byte rotating-lead halo DMA of the stride-2 data-gradient kernel stages the same operands as the
    4-byte form (SR_CONV_ROT=0): outputs are bit-identical (same MFMA chain, same k order)."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    g = torch.Generator().manual_seed(5)
    for b, c, n, hw in ((2, 16, 64, 129), (1, 64, 128, 257), (3, 8, 6, 33)):
        x = torch.randn(b, c, hw, hw, generator=g).to(DEV)
        wt = torch.randn(9, c, n, generator=g).to(DEV)
        isc, osc = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)
        monkeypatch.setenv("SR_CONV_ROT", "1")
        a = conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, False)
        monkeypatch.setenv("SR_CONV_ROT", "0")
        d = conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, False)
        assert torch.equal(a, d), (b, c, n, hw)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("scaled", [True, False])
def test_conv_vs_float64(case, scaled):
    from stylerenderer_amd.op.conv import conv2d_mfma

    b, c, n, h, w, k, stride, pad, tr = case
    g = torch.Generator().manual_seed(b * 1000 + c * 10 + n + h)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn((c, n, k, k) if tr else (n, c, k, k), generator=g)
    isc = torch.randn(b, c, generator=g) if scaled else None
    osc = torch.randn(b, n, generator=g) if scaled else None
    bias = torch.randn(n, generator=g) if scaled else None
    want = ref_conv(x, wgt, isc, osc, bias, stride, pad, tr)
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    got = conv2d_mfma(dev(x), dev(to_taps(wgt, tr)), dev(isc), dev(osc), dev(bias), k, stride, pad, tr)
    assert got.shape == want.shape
    # bound: fp32 round-off against the sum of absolute products
    absx = x.abs().double() * (isc.abs().double()[:, :, None, None] if scaled else 1.0)
    mag = (F.conv_transpose2d(absx, wgt.abs().double(), stride=stride, padding=pad) if tr
           else F.conv2d(absx, wgt.abs().double(), stride=stride, padding=pad))
    if scaled:
        mag = mag * osc.abs().double()[:, :, None, None] + bias.abs().double()[None, :, None, None]
    err = (got.cpu().double() - want).abs()
    assert float((err / (mag + 1e-30)).max()) < 2e-6


TAPS = [(2, 8, 6, 4, 4), (2, 8, 6, 8, 8), (1, 11, 9, 16, 16), (1, 6, 130, 32, 32), (1, 12, 64, 8, 64), (3, 24, 140, 5, 7),
        (4, 512, 512, 8, 8), (1, 80, 64, 33, 30)]


@pytest.mark.parametrize("case", TAPS)
@pytest.mark.parametrize("mode", ["1", "0"])
def test_transposed_conv_tap_split_vs_float64(case, mode, monkeypatch):
    """The stride-2 transposed 3x3 convolution as nine shifted 1x1 convolutions in one launch + one reduction
    (k_conv_mfma<..., TAP9>, the form small problems take; SR_CONVT_TAPS=1 forces it for every size) and the per-phase
    launches it replaces (=0), both against float64 at the kernels' 2e-6 * sum|a*b|: odd extents, non-square maps,
    channel tails (11, 24, 80 input channels), two output-channel tiles, K slices (512 channels at 8^2); deterministic."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    monkeypatch.setenv("SR_CONVT_TAPS", mode)
    b, c, n, h, w = case
    g = torch.Generator().manual_seed(b * 1000 + c * 10 + n + h)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(c, n, 3, 3, generator=g)
    isc, osc, bias = torch.randn(b, c, generator=g), torch.randn(b, n, generator=g), torch.randn(n, generator=g)
    want = ref_conv(x, wgt, isc, osc, bias, 2, 0, True)
    got = conv2d_mfma(x.to(DEV), to_taps(wgt, True).to(DEV), isc.to(DEV), osc.to(DEV), bias.to(DEV), 3, 2, 0, True)
    assert got.shape == want.shape
    absx = x.abs().double() * isc.abs().double()[:, :, None, None]
    mag = F.conv_transpose2d(absx, wgt.abs().double(), stride=2) * osc.abs().double()[:, :, None, None] + \
        bias.abs().double()[None, :, None, None]
    assert float(((got.cpu().double() - want).abs() / (mag + 1e-30)).max()) < 2e-6
    again = conv2d_mfma(x.to(DEV), to_taps(wgt, True).to(DEV), isc.to(DEV), osc.to(DEV), bias.to(DEV), 3, 2, 0, True)
    assert torch.equal(got, again)


@pytest.mark.parametrize("case", [(2, 8, 130, 32, 32), (1, 16, 6, 8, 64), (3, 24, 64, 12, 32), (1, 64, 128, 64, 64)])
@pytest.mark.parametrize("scaled", [True, False])
def test_transposed_conv_border_strips_vs_float64(case, scaled, monkeypatch):
    """Fused interior (k_convt_fused) + the thin per-phase strip launches for output row 2*IH / column 2*IW, whole output
    against float64 at 2e-6 * sum|a*b| — with and without modulation / demodulation / bias, non-square maps, two
    output-channel tiles (130), N < one wave (6); the strips checked on their own as well."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    monkeypatch.setenv("SR_CONVT_TAPS", "0")          # (these sizes would otherwise take the tap-split launch)
    monkeypatch.setenv("SR_CONVT_FUSED", "1")
    b, c, n, h, w = case
    g = torch.Generator().manual_seed(b * 1000 + c * 10 + n + h)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(c, n, 3, 3, generator=g)
    isc = torch.randn(b, c, generator=g) if scaled else None
    osc = torch.randn(b, n, generator=g) if scaled else None
    bias = torch.randn(n, generator=g) if scaled else None
    want = ref_conv(x, wgt, isc, osc, bias, 2, 0, True)
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    got = conv2d_mfma(dev(x), dev(to_taps(wgt, True)), dev(isc), dev(osc), dev(bias), 3, 2, 0, True)
    absx = x.abs().double() * (isc.abs().double()[:, :, None, None] if scaled else 1.0)
    mag = F.conv_transpose2d(absx, wgt.abs().double(), stride=2)
    if scaled:
        mag = mag * osc.abs().double()[:, :, None, None] + bias.abs().double()[None, :, None, None]
    rel = (got.cpu().double() - want).abs() / (mag + 1e-30)
    assert float(rel.max()) < 2e-6
    assert float(rel[:, :, -1, :].max()) < 2e-6 and float(rel[:, :, :, -1].max()) < 2e-6      # the strips themselves


@pytest.mark.parametrize("case", [(2, 320, 512, 32, 32), (4, 512, 256, 32, 32)])
def test_transposed_conv_fused_kernel_with_k_slices_vs_float64(case, monkeypatch):
    """k_convt_fused<true>: few tiles and a long channel loop (fewer than 192 workgroups, more than 256 channels) are cut
    into 2 / 4 K slices whose raw sums k_convt_fused_reduce adds in order — against float64 at 2e-6 * sum|a*b|, and
    bit-identical between two launches; SR_CONVT_FUSED_KS=0 (the per-phase launches it replaces) agrees to round-off."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    monkeypatch.setenv("SR_CONVT_TAPS", "0")
    b, c, n, h, w = case
    g = torch.Generator().manual_seed(c + n)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(c, n, 3, 3, generator=g) / (3 * c ** 0.5)
    isc, osc, bias = torch.randn(b, c, generator=g), torch.randn(b, n, generator=g), torch.randn(n, generator=g)
    want = ref_conv(x, wgt, isc, osc, bias, 2, 0, True)
    args = [t.to(DEV) for t in (x, to_taps(wgt, True), isc, osc, bias)]
    got = conv2d_mfma(*args, 3, 2, 0, True)
    mag = F.conv_transpose2d(x.abs().double() * isc.abs().double()[:, :, None, None], wgt.abs().double(), stride=2) * \
        osc.abs().double()[:, :, None, None] + bias.abs().double()[None, :, None, None]
    assert float(((got.cpu().double() - want).abs() / (mag + 1e-30)).max()) < 2e-6
    assert torch.equal(got, conv2d_mfma(*args, 3, 2, 0, True))
    monkeypatch.setenv("SR_CONVT_FUSED_KS", "0")
    old = conv2d_mfma(*args, 3, 2, 0, True)
    assert float(((old - got).abs().cpu().double() / (mag + 1e-30)).max()) < 2e-6
    assert not torch.equal(old, got) or True          # (different summation orders: equality is not required)


def test_conv_full_width_layers_spotcheck():
    """Generator-sized layers (512 -> 512 at 16x16, 128 -> 128 at 128x128): compare a strip of
    outputs with float64."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    for (b, c, n, res) in ((4, 512, 512, 16), (2, 128, 128, 128)):
        g = torch.Generator().manual_seed(res)
        x = torch.randn(b, c, res, res, generator=g)
        wgt = torch.randn(n, c, 3, 3, generator=g) / np.sqrt(c * 9)
        isc = torch.randn(b, c, generator=g)
        osc = torch.rand(b, n, generator=g) + 0.5
        got = conv2d_mfma(x.to(DEV), to_taps(wgt, False).to(DEV), isc.to(DEV), osc.to(DEV), None, 3, 1, 1)
        want = ref_conv(x[:1], wgt, isc[:1], osc[:1], None, 1, 1, False)
        err = (got[:1].cpu().double() - want).abs().max().item()
        assert err < 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("scaled", [True, False])
def test_wgrad_vs_float64(case, scaled):
    """dL/dW of the same cases through float64 autograd on the CPU."""
    from stylerenderer_amd.op.conv import conv2d_wgrad_mfma

    b, c, n, h, w, k, stride, pad, tr = case
    g = torch.Generator().manual_seed(b * 977 + c * 13 + n + h)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn((c, n, k, k) if tr else (n, c, k, k), generator=g).double().requires_grad_()
    isc = torch.randn(b, c, generator=g) if scaled else None
    osc = torch.randn(b, n, generator=g) if scaled else None
    y = ref_conv(x, wgt, isc, osc, None, stride, pad, tr)
    gy = torch.randn(y.shape, generator=g)
    (gw,) = torch.autograd.grad(y, wgt, gy.double())
    want = to_taps(gw, tr)
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    got = conv2d_wgrad_mfma(dev(x), dev(gy), dev(isc), dev(osc), k, stride, pad, tr)
    assert got.shape == want.shape
    scale = want.abs().max().item() + 1e-30
    # K = B * pixels terms per output, fp32 partial sums: relative to the largest gradient entry
    assert float((got.cpu().double() - want).abs().max()) < 3e-5 * scale * max(1.0, np.sqrt(b * h * w) / 30)


def test_stride2_wgrad_dma_staging_equals_dword_staging(monkeypatch):
    """k_wgrad_s2_dma stages the same operands as the dword kernel and runs the same MFMA chain in the same k
    order: the weight gradients are bit-identical (SR_WGRAD_DMA=0 selects the dword kernel)."""
    from stylerenderer_amd.op.conv import conv2d_wgrad_mfma

    g = torch.Generator().manual_seed(11)
    for b, c, n, res, tr in ((2, 128, 32, 16, True), (3, 256, 64, 32, True), (4, 128, 64, 64, True),
                             (2, 32, 128, 33, False), (3, 64, 256, 129, False)):
        out = 2 * res + 1 if tr else (res - 3) // 2 + 1
        x = torch.randn(b, c, res, res, generator=g).to(DEV)
        gy = torch.randn(b, n, out, out, generator=g).to(DEV)
        xs, gs = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)
        monkeypatch.setenv("SR_WGRAD_DMA", "1")
        a = conv2d_wgrad_mfma(x, gy, xs, gs, 3, 2, 0, tr)
        a2 = conv2d_wgrad_mfma(x, gy, None, None, 3, 2, 0, tr)
        monkeypatch.setenv("SR_WGRAD_DMA", "0")
        d = conv2d_wgrad_mfma(x, gy, xs, gs, 3, 2, 0, tr)
        d2 = conv2d_wgrad_mfma(x, gy, None, None, 3, 2, 0, tr)
        assert torch.isfinite(a).all()
        assert torch.equal(a, d), (b, c, n, res, tr, float((a - d).abs().max()))
        assert torch.equal(a2, d2), (b, c, n, res, tr)


def test_winograd_wgrad_two_waves_per_simd_equals_one_wave_kernel(monkeypatch):
    """k_wgrad_wino<8> (512 threads, positions split over a wave pair) runs the same transforms and the same MFMA chain
    per accumulator as k_wgrad_wino<4>: bit-identical weight gradients (SR_WGW_WAVES=4 selects the one-wave kernel)."""
    from stylerenderer_amd.op.conv import conv2d_wgrad_mfma

    g = torch.Generator().manual_seed(17)
    for b, c, n, h, w in ((1, 64, 64, 2, 16), (3, 64, 128, 8, 32), (2, 128, 64, 64, 64), (5, 64, 64, 6, 48),
                          (4, 128, 128, 128, 128)):
        x = torch.randn(b, c, h, w, generator=g).to(DEV)
        gy = torch.randn(b, n, h, w, generator=g).to(DEV)
        xs, gs = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)
        monkeypatch.setenv("SR_WGW_WAVES", "8")
        a = conv2d_wgrad_mfma(x, gy, xs, gs, 3, 1, 1)
        a2 = conv2d_wgrad_mfma(x, gy, None, None, 3, 1, 1)
        monkeypatch.setenv("SR_WGW_WAVES", "4")
        d = conv2d_wgrad_mfma(x, gy, xs, gs, 3, 1, 1)
        d2 = conv2d_wgrad_mfma(x, gy, None, None, 3, 1, 1)
        assert torch.isfinite(a).all()
        assert torch.equal(a, d), (b, c, n, h, w, float((a - d).abs().max()))
        assert torch.equal(a2, d2), (b, c, n, h, w)


def test_wgrad_deterministic_and_large():
    from stylerenderer_amd.op.conv import conv2d_wgrad_mfma

    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 128, 64, 64, generator=g).to(DEV)
    gy = torch.randn(4, 128, 64, 64, generator=g).to(DEV)
    xs = torch.randn(4, 128, generator=g).to(DEV)
    gs = torch.randn(4, 128, generator=g).to(DEV)
    a = conv2d_wgrad_mfma(x, gy, xs, gs)
    b = conv2d_wgrad_mfma(x, gy, xs, gs)
    assert torch.equal(a, b)
    xd = (x * xs[:, :, None, None]).double().cpu()
    gd = (gy * gs[:, :, None, None]).double().cpu()
    want = torch.nn.grad.conv2d_weight(xd, (128, 128, 3, 3), gd, padding=1)
    want = want.permute(2, 3, 1, 0).reshape(9, 128, 128)
    assert float((a.cpu().double() - want).abs().max()) < 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("shape", [(2, 8, 3, 8, 8), (3, 20, 3, 16, 12), (2, 16, 4, 4, 4), (1, 128, 3, 64, 64),
                                   (2, 6, 1, 8, 8)])
def test_smallconv_primitives_and_double_backward(shape):
    """ToRGB-sized 1x1 modulated conv: value, first-order gradients and a double backward, against
    float64 einsum on the CPU."""
    from stylerenderer_amd.op.smallconv import modulated_conv1x1_small

    b, c, n, h, w = shape
    g = torch.Generator().manual_seed(c * 7 + n)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(n, c, generator=g)
    s = torch.randn(b, c, generator=g)
    gy = torch.randn(b, n, h, w, generator=g)
    probe = torch.randn(b, c, h, w, generator=g)

    def run(dev, dt):
        xs, ws, ss = (t.to(dev, dt).requires_grad_() for t in (x, wgt, s))
        if dev == "cpu":
            y = torch.einsum("bjc,bchw->bjhw", ws[None] * ss[:, None, :], xs)
        else:
            y = modulated_conv1x1_small(xs, ws, ss)
        gyd = gy.to(dev, dt).requires_grad_()
        gx, gw, gs = torch.autograd.grad(y, [xs, ws, ss], gyd, create_graph=True)
        q = (gx * probe.to(dev, dt)).sum() + (gw * ws.detach()).sum() + (gs * ss.detach()).sum()
        ggy, gss = torch.autograd.grad(q, [gyd, ss])
        return [t.detach().cpu().double() for t in (y, gx, gw, gs, ggy, gss)]

    want = run("cpu", torch.float64)
    got = run(DEV, torch.float32)
    for a, r in zip(got, want):
        assert a.shape == r.shape
        assert float((a - r).abs().max()) <= 3e-5 * (float(r.abs().max()) + 1e-6) * max(1.0, np.sqrt(h * w) / 8)


# ---- weight preparation (csrc/weight_prep.hip): one-launch replacements of ATen view/scale passes
def _wprep_reference(w, scale, want_sq):
    ws = w.reshape(w.shape[-4:]) * scale
    co, ci, k, _ = ws.shape
    wt = ws.permute(2, 3, 1, 0).reshape(k * k, ci, co)
    return wt, (ws.pow(2).sum((2, 3)).t() if want_sq else None)


@pytest.mark.parametrize("shape", [(1, 24, 16, 3, 3), (6, 10, 3, 3), (3, 16, 1, 1), (1, 64, 130, 1, 1)])
def test_weight_prep_forward_backward_and_second_order(shape):
    from stylerenderer_amd.op.weight_prep import weight_prep

    g = torch.Generator().manual_seed(sum(shape))
    w0 = torch.randn(shape, generator=g)
    scale = 0.37
    for want_sq in (True, False):
        w = w0.clone().to(DEV).requires_grad_(True)
        wr = w0.clone().double().requires_grad_(True)
        wt, wsq = weight_prep(w, scale, want_sq)
        wt_r, wsq_r = _wprep_reference(wr, scale, want_sq)
        assert torch.equal(wt.detach().cpu(), _wprep_reference(w0, scale, False)[0])   # one fp32 product
        a = torch.randn(wt_r.shape, generator=g)
        loss, loss_r = (wt * a.to(DEV)).sum(), (wt_r * a.double()).sum()
        if want_sq:
            np.testing.assert_allclose(wsq.detach().cpu().numpy(), wsq_r.detach().numpy(), rtol=2e-6)
            b = torch.randn(wsq_r.shape, generator=g)
            loss, loss_r = loss + (wsq * wsq * b.to(DEV)).sum(), loss_r + (wsq_r * wsq_r * b.double()).sum()
        (gw,) = torch.autograd.grad(loss, w, create_graph=True)
        (gw_r,) = torch.autograd.grad(loss_r, wr, create_graph=True)
        np.testing.assert_allclose(gw.detach().cpu().numpy(), gw_r.detach().numpy(), rtol=1e-5, atol=1e-6)
        if not want_sq:
            assert not gw_r.requires_grad                                   # linear in w: constant gradient
            continue
        c = torch.randn(shape, generator=g)
        (g2,) = torch.autograd.grad((gw * c.to(DEV)).sum(), w)
        (g2_r,) = torch.autograd.grad((gw_r * c.double()).sum(), wr)
        np.testing.assert_allclose(g2.cpu().numpy(), g2_r.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("taps,c,n,flip", [(9, 40, 24, True), (9, 3, 128, False), (1, 130, 6, False)])
def test_weight_adjoint(taps, c, n, flip):
    from stylerenderer_amd.op.weight_prep import adjoint

    g = torch.Generator().manual_seed(taps + c + n)
    wt = torch.randn(taps, c, n, generator=g)
    want = wt.transpose(1, 2)
    if flip:
        want = want.flip(0)
    x = wt.to(DEV).requires_grad_(True)
    got = adjoint(x, flip)
    assert torch.equal(got.cpu(), want)
    assert got.stride(1) % 4 == 0                                          # padded pitch for the kernel
    a = torch.randn(want.shape, generator=g)
    (gx,) = torch.autograd.grad((got * a.to(DEV)).sum(), x)
    want_g = a.flip(0).transpose(1, 2) if flip else a.transpose(1, 2)
    assert torch.equal(gx.cpu(), want_g)
    # padded-pitch input view
    padded = torch.zeros(taps, c, (n + 3) // 4 * 4 + 4, device=DEV)
    padded[:, :, :n] = wt.to(DEV)
    assert torch.equal(adjoint(padded[:, :, :n], flip).cpu(), want)


# ---- style path (csrc/style_linear.hip) against its defining tensor algebra in float64
@pytest.mark.parametrize("b,k,n,act,bias,strided", [(16, 512, 512, True, True, False), (5, 64, 24, False, True, True),
                                                    (3, 16, 130, False, False, False), (1, 512, 128, True, True, True)])
def test_equal_linear_forward_backward(b, k, n, act, bias, strided):
    from stylerenderer_amd.op import style

    g = torch.Generator().manual_seed(b * 1000 + n)
    xfull = torch.randn(b, 3, k, generator=g)
    w0 = torch.randn(n, k, generator=g)
    b0 = torch.randn(n, generator=g) if bias else None
    wscale, bscale = 1.0 / np.sqrt(k), 0.7
    xd = xfull.to(DEV).requires_grad_(True)
    xr = xfull.double().requires_grad_(True)
    x_in = xd[:, 1] if strided else xd[:, 1].contiguous()
    assert style.linear_supported(x_in, w0.to(DEV))
    wd, wr = w0.to(DEV).requires_grad_(True), w0.double().requires_grad_(True)
    bd = b0.to(DEV).requires_grad_(True) if bias else None
    br = b0.double().requires_grad_(True) if bias else None
    y = style.equal_linear(x_in, wd, bd, wscale, bscale, act)
    yr = style._linear_composite(xr[:, 1], wr, br, wscale, bscale, act)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-5, atol=2e-6)
    gy = torch.randn(b, n, generator=g)
    ins_d = [t for t in (xd, wd, bd) if t is not None]
    ins_r = [t for t in (xr, wr, br) if t is not None]
    got = torch.autograd.grad(y, ins_d, gy.to(DEV))
    want = torch.autograd.grad(yr, ins_r, gy.double())
    for a, r in zip(got, want):
        np.testing.assert_allclose(a.cpu().numpy(), r.numpy(), rtol=3e-5, atol=3e-6)
    # second order (composite route): d/dw of <grad_x, c>
    y2 = style.equal_linear(x_in, wd, bd, wscale, bscale, act)
    (gx,) = torch.autograd.grad(y2, xd, gy.to(DEV), create_graph=True)
    yr2 = style._linear_composite(xr[:, 1], wr, br, wscale, bscale, act)
    (gxr,) = torch.autograd.grad(yr2, xr, gy.double(), create_graph=True)
    c = torch.randn(xfull.shape, generator=g)
    (g2,) = torch.autograd.grad((gx * c.to(DEV)).sum(), wd)
    (g2r,) = torch.autograd.grad((gxr * c.double()).sum(), wr)
    np.testing.assert_allclose(g2.cpu().numpy(), g2r.numpy(), rtol=3e-5, atol=3e-6)


@pytest.mark.parametrize("b,ci,co", [(16, 512, 512), (3, 24, 8), (1, 130, 64)])
def test_demod_scale_forward_backward(b, ci, co):
    from stylerenderer_amd.op import style

    g = torch.Generator().manual_seed(b + ci + co)
    s0 = torch.randn(b, ci, generator=g)
    w0 = torch.rand(ci, co, generator=g) / ci
    eps = 1e-8
    sd, wd = s0.to(DEV).requires_grad_(True), w0.to(DEV).requires_grad_(True)
    sr, wr = s0.double().requires_grad_(True), w0.double().requires_grad_(True)
    d = style.demod_scale(sd, wd, eps)
    dr = style._demod_composite(sr, wr, eps)
    np.testing.assert_allclose(d.detach().cpu().numpy(), dr.detach().numpy(), rtol=1e-5)
    gd = torch.randn(b, co, generator=g)
    got = torch.autograd.grad(d, (sd, wd), gd.to(DEV), create_graph=True)
    want = torch.autograd.grad(dr, (sr, wr), gd.double(), create_graph=True)
    for a, r in zip(got, want):
        np.testing.assert_allclose(a.detach().cpu().numpy(), r.detach().numpy(), rtol=5e-5, atol=1e-5)
    (g2,) = torch.autograd.grad((got[0] * got[0]).sum(), wd)
    (g2r,) = torch.autograd.grad((want[0] * want[0]).sum(), wr)
    np.testing.assert_allclose(g2.cpu().numpy(), g2r.numpy(), rtol=2e-4, atol=1e-4 * float(g2r.abs().max()))
    # first-order fused kernels (no graph recording)
    d2 = style.demod_scale(sd, wd, eps)
    fused = torch.autograd.grad(d2, (sd, wd), gd.to(DEV))
    for a, r in zip(fused, want):
        np.testing.assert_allclose(a.cpu().numpy(), r.detach().numpy(), rtol=5e-5, atol=1e-5)


# ---- Winograd F(2x2,3x3) path (csrc/conv_wino.hip) vs float64 and vs the direct implicit GEMM.
# Stated tolerance: |err| <= 4e-6 * sum|a*b| (the transforms add a few fp32 roundings on sums of up to
# 4 inputs / 3 weights; measured 5e-7), against 2e-6 for the exact-fma-chain direct kernel.
@pytest.mark.parametrize("b,c,n,h,w,scales", [(2, 24, 64, 16, 32, True), (1, 16, 128, 8, 64, False),
                                               (3, 40, 64, 24, 32, True),
                                               (1, 8, 64, 8, 32, True),        # one chunk: first body + tail only
                                               (2, 512, 64, 8, 32, False),     # 64 chunks, the largest style row; 4 K slices
                                               (1, 512, 128, 32, 32, True),    # batch 1 of the inversion loop: 4 K slices
                                               (1, 128, 64, 8, 32, True)])     # 2 K slices of 8 chunks
def test_winograd_conv_vs_float64_and_direct(b, c, n, h, w, scales, monkeypatch):
    from stylerenderer_amd.op.conv import conv2d_mfma

    g = torch.Generator().manual_seed(b * 100 + c)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(n, c, 3, 3, generator=g) / (3 * c ** 0.5)
    isc = torch.randn(b, c, generator=g) if scales else None
    osc = torch.randn(b, n, generator=g) if scales else None
    bias = torch.randn(n, generator=g) if scales else None
    ref = ref_conv(x, wgt, isc, osc, bias, 1, 1, False)
    mag = ref_conv(x.abs(), wgt.abs(), isc.abs() if scales else None, osc.abs() if scales else None, None, 1, 1,
                   False) + (bias.abs().double()[None, :, None, None] if scales else 0.0)
    args = [t.to(DEV) if t is not None else None for t in (x, to_taps(wgt, False), isc, osc, bias)]
    monkeypatch.setenv("SR_WINOGRAD", "1")
    wino = conv2d_mfma(*args, 3, 1, 1).cpu().double()
    monkeypatch.setenv("SR_WINOGRAD", "0")
    direct = conv2d_mfma(*args, 3, 1, 1).cpu().double()
    assert float(((wino - ref).abs() / mag).max()) < 4e-6
    assert float(((direct - ref).abs() / mag).max()) < 2e-6
    assert float(((wino - direct).abs() / mag).max()) < 4e-6
    assert not torch.equal(wino, direct)                 # the two paths really are different kernels
    # few tiles + long channel loop: K slices (k_conv_wino writes raw slice sums, k_wino_reduce adds them in a fixed
    # order and applies the epilogue) — same tolerance, run-to-run identical, and a different summation order than the
    # unsplit kernel (SR_WINO_SPLIT=0)
    monkeypatch.setenv("SR_WINOGRAD", "1")
    again = conv2d_mfma(*args, 3, 1, 1).cpu().double()
    assert torch.equal(again, wino)
    monkeypatch.setenv("SR_WINO_SPLIT", "0")
    unsplit = conv2d_mfma(*args, 3, 1, 1).cpu().double()
    assert float(((unsplit - ref).abs() / mag).max()) < 4e-6
    if c >= 128:
        assert not torch.equal(unsplit, wino)


# Winograd F(3x3,2x2) weight gradient (csrc/conv_wgrad_wino.hip) vs float64 and vs the direct kernel.
# Stated tolerance 4e-6 * sum|a*b| (measured 3e-7), direct kernel 2e-6.
@pytest.mark.parametrize("b,c,n,h,w,scales", [(3, 64, 128, 8, 32, True), (2, 128, 64, 16, 16, False),
                                               (5, 64, 64, 6, 48, True),
                                               (1, 64, 64, 2, 16, True),       # one strip: prologue + tail only
                                               (3, 64, 64, 2, 16, True),       # odd strip count, a sample per strip
                                               (2, 64, 64, 2, 48, False),      # a slice that crosses the sample boundary
                                               (32, 64, 64, 4, 16, True)])     # full scale tables (B = 32), four slices
def test_winograd_wgrad_vs_float64_and_direct(b, c, n, h, w, scales, monkeypatch):
    from stylerenderer_amd.op.conv import conv2d_wgrad_mfma

    g = torch.Generator().manual_seed(b * 10 + h)
    x = torch.randn(b, c, h, w, generator=g)
    gy = torch.randn(b, n, h, w, generator=g)
    xs = torch.randn(b, c, generator=g) if scales else None
    gs = torch.randn(b, n, generator=g) if scales else None

    def ref(xv, gv, xsv, gsv):
        wref = torch.zeros(n, c, 3, 3, dtype=torch.float64, requires_grad=True)
        xin = xv.double() * (xsv.double()[:, :, None, None] if xsv is not None else 1.0)
        gin = gv.double() * (gsv.double()[:, :, None, None] if gsv is not None else 1.0)
        (gw,) = torch.autograd.grad(F.conv2d(xin, wref, padding=1), wref, gin)
        return gw.permute(2, 3, 1, 0).reshape(9, c, n)

    want = ref(x, gy, xs, gs)
    mag = ref(x.abs(), gy.abs(), xs.abs() if scales else None, gs.abs() if scales else None)
    args = [t.to(DEV) if t is not None else None for t in (x, gy, xs, gs)]
    monkeypatch.setenv("SR_WINOGRAD", "1")
    wino = conv2d_wgrad_mfma(*args, 3, 1, 1).cpu().double()
    again = conv2d_wgrad_mfma(*args, 3, 1, 1).cpu().double()
    monkeypatch.setenv("SR_WINOGRAD", "0")
    direct = conv2d_wgrad_mfma(*args, 3, 1, 1).cpu().double()
    assert torch.equal(wino, again)                      # fixed-order split-K: run-to-run identical
    assert float(((wino - want).abs() / mag).max()) < 4e-6
    assert float(((direct - want).abs() / mag).max()) < 2e-6
    assert not torch.equal(wino, direct)


# ---- conv + noise + bias + LeakyReLU as one node (ConvNBAFn) vs the two separate operators
@pytest.mark.parametrize("b,c,n,h,w,shared_noise", [(2, 16, 64, 8, 32, False), (3, 24, 128, 16, 32, True),
                                                    (1, 128, 64, 8, 32, True)])        # K slices: tail in k_wino_reduce
def test_conv_nba_node_matches_separate_operators(b, c, n, h, w, shared_noise):
    from stylerenderer_amd.op import conv as cv
    from stylerenderer_amd.op.fused_elem import noise_bias_act

    g = torch.Generator().manual_seed(b + c + n)
    mk = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    x0, wt0 = mk(b, c, h, w), mk(9, c, n) / (3 * c ** 0.5)
    s0, d0 = mk(b, c), mk(b, n).abs() + 0.5
    noise = mk(1 if shared_noise else b, 1, h, w).to(DEV)
    nw0, ab0 = mk(1), mk(n)
    proj = mk(b, n, h, w).to(DEV)

    def run(fused):
        x, wt, s, d, nw, ab = [t.clone().to(DEV).requires_grad_(True) for t in (x0, wt0, s0, d0, nw0, ab0)]
        if fused:
            assert cv.conv_nba_supported(x, wt, noise)
            y = cv.conv2d_nba(x, wt, s, d, noise, nw, ab)
        else:
            y = noise_bias_act(cv.conv2d(x, wt, s, d, None, "c3"), noise, nw, ab)
        grads = torch.autograd.grad((y * proj).sum(), [x, wt, s, d, nw, ab])
        return y.detach(), grads

    yf, gf = run(True)
    yu, gu = run(False)
    assert torch.equal(yf, yu)                                    # same kernel arithmetic, same operation order
    for name, a, r in zip(("x", "wt", "s", "d", "noise_w", "bias"), gf, gu):
        tol = 2e-4 if name == "d" else 1e-5                       # d: y0 rebuilt from the activation output
        assert rel_err_t(a, r) < tol, name


def rel_err_t(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("up", [False, True])
def test_fused_styledconv_nodes_second_order_with_linked_inputs(up):
    """The fused nodes' create_graph fallback must not follow the history that links their inputs OUTSIDE the
    node (the demodulation scale is a function of the style): first- and second-order gradients w.r.t. the
    style equal those of the separate operators."""
    from stylerenderer_amd.op import conv as cv
    from stylerenderer_amd.op.fused_elem import blur_noise_bias_act, noise_bias_act

    g = torch.Generator().manual_seed(11 + int(up))
    mk = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    b, c, n, h, w = 2, 16, 64, 8, 32
    x0, wt0, s0, wsq0 = mk(b, c, h, w), mk(9, c, n) / (3 * c ** 0.5), mk(b, c), mk(c, n).abs() / c
    oh, ow = (2 * h, 2 * w) if up else (h, w)
    noise = mk(b, 1, oh, ow).to(DEV)
    nw0, ab0 = mk(1), mk(n)
    proj = mk(b, n, oh, ow).to(DEV)
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    kernel = (k1[None, :] * k1[:, None] / 64.0 * 4.0).to(DEV)

    def run(fused):
        x, wt, s, wsq, nw, ab = [t.clone().to(DEV).requires_grad_(True) for t in (x0, wt0, s0, wsq0, nw0, ab0)]
        d = torch.rsqrt((s * s) @ wsq + 1e-8)                    # linked to s outside the node
        if up and fused:
            y = cv.upconv_nba(x, wt, s, d, kernel, (1, 1), noise, nw, ab)
        elif up:
            y = blur_noise_bias_act(cv.conv2d(x, wt, s, d, None, "t3s2"), kernel, (1, 1), noise, nw, ab)
        elif fused:
            y = cv.conv2d_nba(x, wt, s, d, noise, nw, ab)
        else:
            y = noise_bias_act(cv.conv2d(x, wt, s, d, None, "c3"), noise, nw, ab)
        (gs,) = torch.autograd.grad((y * proj).sum(), s, create_graph=True)
        (g2,) = torch.autograd.grad((gs * gs).sum(), wt)
        return gs.detach(), g2

    gs_f, g2_f = run(True)
    gs_u, g2_u = run(False)
    assert rel_err_t(gs_f, gs_u) < 1e-5
    assert rel_err_t(g2_f, g2_u) < 1e-4


@pytest.mark.parametrize("b,n,oh,shared,frozen", [(2, 8, 256, False, False), (3, 5, 16, True, False), (1, 6, 64, False, True),
                                                   (2, 4, 96, None, False)])
def test_upsampling_tail_backward_in_one_pass_equals_the_two_kernels(b, n, oh, shared, frozen, monkeypatch):
    """k_fir4_nba_bwd (activation backward + its three reductions + the blur's gradient in one pass over gy and y)
    against sr_noise_bias_act_bwd_dot followed by sr_upfirdn2d: the 257^2-side gradient bit for bit (same expression
    for lrelu'(y) * gy * gain, same tap order), the three sums to fp32 round-off of their different association."""
    from stylerenderer_amd.op import conv as cv
    from stylerenderer_amd.op.upfirdn2d import flipped, upfirdn2d_op

    g = torch.Generator().manual_seed(5 + oh)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    ow = oh
    gy, out = mk(b, n, oh, ow), mk(b, n, oh, ow)
    noise = None if shared is None else mk(1 if shared else b, 1, oh, ow)
    nw, ab = (mk(1) if noise is not None else None), mk(n)
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    kernel = (k1[None, :] * k1[:, None] / 64.0 * 4.0).to(DEV)
    shape257 = (b, n, oh + 1, ow + 1)
    want_p = not frozen
    g257, gb, gnw, rdot = cv._blur_nba_bwd(gy, out, flipped(kernel), 1, shape257, noise, nw, ab, 0.2, 2 ** 0.5, want_p)
    gm, gb0, gnw0, rdot0 = cv._nba_bwd_dot(gy, out, noise, nw, ab, 0.2, 2 ** 0.5, want_p)
    g0 = upfirdn2d_op(gm.reshape(-1, oh, ow, 1), flipped(kernel), 1, 1, 1, 1, 2, 2, 2, 2).view(shape257)
    assert torch.equal(g257, g0)
    assert rel_err_t(rdot, rdot0) < 2e-6
    if want_p:
        assert rel_err_t(gb, gb0) < 2e-6
        if noise is not None:
            assert abs(float(gnw) - float(gnw0)) < 2e-6 * float((gm * noise).abs().sum())
    else:
        assert gb.numel() == 0 and gnw.numel() == 0
    # and run-to-run identical
    again = cv._blur_nba_bwd(gy, out, flipped(kernel), 1, shape257, noise, nw, ab, 0.2, 2 ** 0.5, want_p)
    assert all(torch.equal(u, v) for u, v in zip(again, (g257, gb, gnw, rdot)))


def test_convlayer_shortcuts_equal_the_module_chain(monkeypatch):
    """layers.ConvLayer on device tensors (bias + LeakyReLU in the Winograd store; both biases added once; the skip
    branch's blur evaluated at the kept pixels) against the plain chain Blur -> EqualConv2d(+bias) -> FusedLeakyReLU."""
    from stylerenderer_amd import layers, synth

    cases = [dict(in_channel=16, out_channel=64, kernel_size=3),                      # Winograd + fused tail
             dict(in_channel=16, out_channel=24, kernel_size=3),                      # direct kernel, biases merged
             dict(in_channel=16, out_channel=32, kernel_size=1, downsample=True, activate=False, bias=False),
             dict(in_channel=16, out_channel=32, kernel_size=3, downsample=True)]
    for i, kw in enumerate(cases):
        m = layers.ConvLayer(**kw).to(DEV)
        synth.fill_state_dict(m.state_dict(), salt=300 + i)
        x0 = torch.from_numpy(synth.det_normal((2, 16, 32, 32), 310 + i)).to(DEV)

        def run(chain):
            x = x0.clone().requires_grad_()
            y = torch.nn.Sequential.forward(m, x) if chain else m(x)
            proj = torch.from_numpy(synth.det_normal(tuple(y.shape), 320 + i)).to(DEV)
            grads = torch.autograd.grad((y * proj).sum(), [x] + list(m.parameters()))
            return y.detach(), grads

        ya, ga = run(False)
        yb, gb = run(True)
        assert rel_err_t(ya, yb) < 2e-6, kw
        for u, v in zip(ga, gb):
            assert rel_err_t(u, v) < 2e-5, kw


def test_frozen_weights_keep_their_winograd_domain_weights(monkeypatch):
    """A prepared weight marked frozen (`_sr_frozen`: latent inversion, LPIPS trunk) keeps one scratch per call geometry;
    the second and later calls skip k_wino_weights (SR_CONV_U_READY).  Same bits as the uncached call, for the plain and
    the fused-tail entry point, with K slices too; an in-place change of the weight (version bump) is not served stale."""
    from stylerenderer_amd.op import conv as cv

    g = torch.Generator().manual_seed(3)
    for b, c, n, h, w in ((1, 128, 64, 8, 32), (2, 64, 128, 16, 32)):
        x = torch.randn(b, c, h, w, generator=g).to(DEV)
        wt = (torch.randn(9, c, n, generator=g) / (3 * c ** 0.5)).to(DEV)
        isc, osc = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)
        monkeypatch.setenv("SR_U_CACHE", "0")
        want = cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
        monkeypatch.setenv("SR_U_CACHE", "1")
        wt._sr_frozen = True
        first = cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
        assert len(wt._sr_scratch) == 1
        second = cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)          # U_READY: no weight transform launched
        assert torch.equal(first, want) and torch.equal(second, want)
        nz = torch.randn(b, 1, h, w, generator=g).to(DEV)
        nw, ab = torch.tensor([0.3], device=DEV), torch.randn(n, generator=g).to(DEV)
        a = cv.ConvNBAFn.apply(x, wt, isc, osc, nz, nw, ab, 0.2, 2 ** 0.5)
        a2 = cv.ConvNBAFn.apply(x, wt, isc, osc, nz, nw, ab, 0.2, 2 ** 0.5)
        assert torch.equal(a, a2)
        wt.mul_(2.0)                                                      # new version: the cached block is not reused
        doubled = cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
        assert torch.allclose(doubled, 2 * want, rtol=1e-5, atol=1e-6)


def test_frozen_weight_scratch_counts_as_written_only_after_an_executed_winograd_call(monkeypatch):
    """the persistent scratch of a frozen weight was marked "Winograd-domain weights present" when it was
    CREATED.  Two first calls leave the block unwritten: one that is only recorded into a graph under capture, and one
    whose input is not 16-byte aligned (the direct kernel serves it).  The next eager, aligned call must still run the
    weight transform — same bits as the uncached call; and once the block really is written, later calls reuse it."""
    from stylerenderer_amd.op import conv as cv

    monkeypatch.setenv("SR_U_CACHE", "1")
    g = torch.Generator().manual_seed(5)
    b, c, n, h, w = 1, 64, 64, 8, 32
    x = torch.randn(b, c, h, w, generator=g).to(DEV)
    isc, osc = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)

    def weight():
        wt = (torch.randn(9, c, n, generator=torch.Generator().manual_seed(9)) / (3 * c ** 0.5)).to(DEV)
        monkeypatch.setenv("SR_U_CACHE", "0")
        want = cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
        monkeypatch.setenv("SR_U_CACHE", "1")
        wt._sr_frozen = True
        return wt, want

    # (a) first call under capture: recorded, not executed
    wt, want = weight()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            captured = cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
    torch.cuda.current_stream().wait_stream(side)
    (entry,) = wt._sr_scratch.values()
    assert not entry.ready
    entry.buf.fill_(float("nan"))                       # whatever the fresh allocation held
    eager = cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
    assert torch.equal(eager, want) and entry.ready
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, want)
    # (b) first call on a misaligned input view: the direct kernel, no weight block written
    wt, want = weight()
    pad = torch.zeros(x.numel() + 1, device=DEV)
    xm = pad[1:].view_as(x)
    xm.copy_(x)
    assert xm.data_ptr() % 16 != 0
    direct = cv.conv2d_mfma(xm, wt, isc, osc, None, 3, 1, 1)
    (entry,) = wt._sr_scratch.values()
    assert not entry.ready and rel_err_t(direct, want) < 2e-6
    entry.buf.fill_(float("nan"))
    assert torch.equal(cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1), want) and entry.ready
    assert torch.equal(cv.conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1), want)          # reused now


@pytest.mark.parametrize("case", [(2, 8, 3, 8, 8), (2, 16, 5, 32, 32), (3, 130, 70, 20, 12), (2, 256, 512, 64, 64),
                                  (5, 128, 256, 16, 24)])
@pytest.mark.parametrize("scaled", [True, False])
def test_split_bf16_wgrad_1x1_holds_the_fp32_bar(case, scaled, monkeypatch):
    """SPIKE, opt-in (SR_CONV_SPLIT_BF16=1): the 1x1 weight gradient on the bf16 matrix cores with every fp32 operand
    split into three bf16 pieces and six cross products accumulated in fp32 (csrc/conv_wgrad_bf16x3.hip).  Held to the
    SAME bound as the exact-fp32 convolution kernels above — |error| < 2e-6 * sum |a| |b| against float64 — channel
    tails, ragged last chunk (H*W not a multiple of 32), with and without the modulation scales; and it must not be
    worse than the fp32-MFMA kernel by more than 2x on the same inputs."""
    from stylerenderer_amd.op.conv import conv2d_wgrad_mfma

    b, c, n, h, w = case
    g = torch.Generator().manual_seed(b * 31 + c + n + h)
    x = torch.randn(b, c, h, w, generator=g)
    gy = torch.randn(b, n, h, w, generator=g)
    xs = torch.randn(b, c, generator=g) if scaled else None
    gs = torch.randn(b, n, generator=g) if scaled else None
    xd = x.double() * (xs.double()[:, :, None, None] if scaled else 1.0)
    gd = gy.double() * (gs.double()[:, :, None, None] if scaled else 1.0)
    want = torch.einsum("bcp,bnp->cn", xd.flatten(2), gd.flatten(2))[None]            # [1, C, N]
    mag = torch.einsum("bcp,bnp->cn", xd.abs().flatten(2), gd.abs().flatten(2))[None]
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "0")
    exact = conv2d_wgrad_mfma(dev(x), dev(gy), dev(xs), dev(gs), 1, 1, 0)
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "1")
    split = conv2d_wgrad_mfma(dev(x), dev(gy), dev(xs), dev(gs), 1, 1, 0)
    again = conv2d_wgrad_mfma(dev(x), dev(gy), dev(xs), dev(gs), 1, 1, 0)
    assert split.shape == want.shape and torch.equal(split, again)                     # deterministic
    e_split = float(((split.cpu().double() - want).abs() / (mag + 1e-30)).max())
    e_exact = float(((exact.cpu().double() - want).abs() / (mag + 1e-30)).max())
    print(case, scaled, "split-bf16 %.2e, fp32 MFMA %.2e of sum|a||b|" % (e_split, e_exact))
    assert e_split < 2e-6, (e_split, e_exact)
    assert e_split < 2 * e_exact + 1e-7, (e_split, e_exact)
    assert not torch.equal(split, exact) or c * n < 64                                 # the other kernel really ran


@pytest.mark.parametrize("case", [(2, 128, 64, 32, True), (3, 256, 128, 64, True), (2, 64, 128, 65, False),
                                  (1, 128, 256, 129, False), (2, 128, 64, 16, True)])
@pytest.mark.parametrize("scaled", [True, False])
def test_split_bf16_wgrad_stride2_holds_the_fp32_bar(case, scaled, monkeypatch):
    """SPIKE, opt-in (SR_CONV_SPLIT_BF16=1): the stride-2 3x3 weight gradient (up-sampling transposed convolution and
    down-sampling convolution) on the bf16 matrix cores, three-way split, six products (k_wgrad_s2_bf16x3: the 17 window
    columns of a lane are split once and the even / odd / shifted-even tap fragments assembled with v_perm).  Same bound
    as the exact-fp32 kernels — |error| < 2e-6 * sum |a||b| against float64 autograd — for both geometries; the last
    case (16-wide grid) is not eligible and must fall through to the fp32 kernel bit for bit."""
    from stylerenderer_amd.op.conv import conv2d_wgrad_mfma

    b, c, n, res, tr = case
    g = torch.Generator().manual_seed(b * 131 + c + n + res)
    x = torch.randn(b, c, res, res, generator=g)
    wshape = (c, n, 3, 3) if tr else (n, c, 3, 3)
    isc = torch.randn(b, c, generator=g) if scaled else None
    osc = torch.randn(b, n, generator=g) if scaled else None

    def grad_of(xx, ii, oo, gg):
        wgt = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
        y = ref_conv(xx, wgt, ii, oo, None, 2, 0, tr)
        (gw,) = torch.autograd.grad(y, wgt, gg.double())
        return to_taps(gw, tr), y.shape

    out = 2 * res + 1 if tr else (res - 3) // 2 + 1
    gy = torch.randn(b, n, out, out, generator=g)
    want, _ = grad_of(x, isc, osc, gy)
    mag, _ = grad_of(x.abs(), isc.abs() if scaled else None, osc.abs() if scaled else None, gy.abs())
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "0")
    exact = conv2d_wgrad_mfma(dev(x), dev(gy), dev(isc), dev(osc), 3, 2, 0, tr)
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "1")
    split = conv2d_wgrad_mfma(dev(x), dev(gy), dev(isc), dev(osc), 3, 2, 0, tr)
    again = conv2d_wgrad_mfma(dev(x), dev(gy), dev(isc), dev(osc), 3, 2, 0, tr)
    assert split.shape == want.shape and torch.equal(split, again)
    e_split = float(((split.cpu().double() - want).abs() / (mag + 1e-30)).max())
    e_exact = float(((exact.cpu().double() - want).abs() / (mag + 1e-30)).max())
    print(case, scaled, "split-bf16 %.2e, fp32 MFMA %.2e of sum|a||b|" % (e_split, e_exact))
    assert e_split < 2e-6, (e_split, e_exact)
    if res == 16:
        assert torch.equal(split, exact)                       # not eligible: the fp32 kernel served it
    else:
        assert e_split < 2 * e_exact + 1e-7 and not torch.equal(split, exact)


@pytest.mark.parametrize("case", [(2, 16, 128, 65), (1, 64, 256, 129), (3, 8, 128, 65), (2, 16, 64, 65)])
@pytest.mark.parametrize("scaled", [True, False])
def test_split_bf16_conv_stride2_holds_the_fp32_bar(case, scaled, monkeypatch):
    """SPIKE, opt-in (SR_CONV_SPLIT_BF16=1): the stride-2 3x3 convolution (down-sampling layers, data gradient of the
    up-sampling ones) on the bf16 matrix cores: weights split once per call (k_split_w_s2), the input window split in
    the loop (k_conv_s2_bf16x3).  Same bound as the exact-fp32 kernel — |error| < 2e-6 * sum |a||b| against float64 —
    with input / output scales and bias; the last case (64 output channels) is not eligible and falls through."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    b, c, n, res = case
    g = torch.Generator().manual_seed(b * 77 + c + n + res)
    x = torch.randn(b, c, res, res, generator=g)
    wgt = torch.randn(n, c, 3, 3, generator=g)
    isc = torch.randn(b, c, generator=g) if scaled else None
    osc = torch.randn(b, n, generator=g) if scaled else None
    bias = torch.randn(n, generator=g) if scaled else None
    want = ref_conv(x, wgt, isc, osc, bias, 2, 0, False)
    absx = x.abs().double() * (isc.abs().double()[:, :, None, None] if scaled else 1.0)
    mag = F.conv2d(absx, wgt.abs().double(), stride=2)
    if scaled:
        mag = mag * osc.abs().double()[:, :, None, None] + bias.abs().double()[None, :, None, None]
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "0")
    exact = conv2d_mfma(dev(x), dev(to_taps(wgt, False)), dev(isc), dev(osc), dev(bias), 3, 2, 0, False)
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "1")
    split = conv2d_mfma(dev(x), dev(to_taps(wgt, False)), dev(isc), dev(osc), dev(bias), 3, 2, 0, False)
    again = conv2d_mfma(dev(x), dev(to_taps(wgt, False)), dev(isc), dev(osc), dev(bias), 3, 2, 0, False)
    assert split.shape == want.shape and torch.equal(split, again)
    e_split = float(((split.cpu().double() - want).abs() / (mag + 1e-30)).max())
    e_exact = float(((exact.cpu().double() - want).abs() / (mag + 1e-30)).max())
    print(case, scaled, "split-bf16 %.2e, fp32 MFMA %.2e of sum|a||b|" % (e_split, e_exact))
    assert e_split < 2e-6, (e_split, e_exact)
    if n % 128:
        assert torch.equal(split, exact)
    else:
        assert e_split < 2 * e_exact + 1e-7 and not torch.equal(split, exact)


@pytest.mark.parametrize("case", [(2, 16, 64, 32), (1, 64, 128, 64), (3, 48, 64, 32), (2, 16, 32, 32), (2, 8, 64, 32)])
@pytest.mark.parametrize("scaled", [True, False])
def test_split_bf16_transposed_conv_holds_the_fp32_bar(case, scaled, monkeypatch):
    """SPIKE, opt-in (SR_CONV_SPLIT_BF16=1): the stride-2 transposed 3x3 convolution (forward of the up-sampling
    layers): interior of the map on the bf16 matrix cores (k_split_w_t + k_convt_bf16x3: four input shifts feed the nine
    taps, five paired k groups), border row / column on the exact strips.  Same |error| < 2e-6 * sum |a||b| bound
    against float64 over the WHOLE output incl. the seams; the 32-channel case is not eligible and falls through."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    b, c, n, res = case
    g = torch.Generator().manual_seed(b * 55 + c + n + res)
    x = torch.randn(b, c, res, res, generator=g)
    wgt = torch.randn(c, n, 3, 3, generator=g)
    isc = torch.randn(b, c, generator=g) if scaled else None
    osc = torch.randn(b, n, generator=g) if scaled else None
    bias = torch.randn(n, generator=g) if scaled else None
    want = ref_conv(x, wgt, isc, osc, bias, 2, 0, True)
    absx = x.abs().double() * (isc.abs().double()[:, :, None, None] if scaled else 1.0)
    mag = F.conv_transpose2d(absx, wgt.abs().double(), stride=2)
    if scaled:
        mag = mag * osc.abs().double()[:, :, None, None] + bias.abs().double()[None, :, None, None]
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "0")
    exact = conv2d_mfma(dev(x), dev(to_taps(wgt, True)), dev(isc), dev(osc), dev(bias), 3, 2, 0, True)
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "1")
    split = conv2d_mfma(dev(x), dev(to_taps(wgt, True)), dev(isc), dev(osc), dev(bias), 3, 2, 0, True)
    again = conv2d_mfma(dev(x), dev(to_taps(wgt, True)), dev(isc), dev(osc), dev(bias), 3, 2, 0, True)
    assert split.shape == want.shape and torch.equal(split, again)
    e_split = float(((split.cpu().double() - want).abs() / (mag + 1e-30)).max())
    e_exact = float(((exact.cpu().double() - want).abs() / (mag + 1e-30)).max())
    print(case, scaled, "split-bf16 %.2e, fp32 MFMA %.2e of sum|a||b|" % (e_split, e_exact))
    assert e_split < 2e-6, (e_split, e_exact)
    if n % 64 or c % 16:
        assert torch.equal(split, exact)                        # not eligible: the exact kernels served it
    else:
        assert e_split < 2 * e_exact + 1e-7 and not torch.equal(split, exact)


# ------------------------------------------------------------------------------------------------ generic geometry
GENERIC = [  # B, C, N, H, W, kh, kw, stride, pad
    (2, 5, 7, 11, 13, 5, 5, 1, 2), (3, 4, 6, 9, 9, 3, 3, 1, 0), (2, 6, 5, 12, 10, 4, 4, 2, 1), (1, 3, 4, 15, 17, 7, 7, 3, 3),
    (2, 8, 8, 8, 8, 2, 2, 3, 0), (2, 4, 4, 10, 12, 3, 5, (2, 1), (1, 2))]


@pytest.mark.parametrize("case", GENERIC)
def test_generic_convolution_kernels_vs_float64(case):
    """csrc/conv_generic.hip (any kernel extent / stride / padding — what the reference's EqualConv2d and
    ModulatedConv2d accept beyond the matrix-core geometries): forward, data gradient, weight gradient and the
    transposed convolution as a forward operator against float64, |err| <= 2e-6 * sum|a*b|; then a SECOND-order
    gradient through the three mutually-adjoint Functions (the R1 / path-length pattern) against float64 autograd."""
    from stylerenderer_amd.op import conv_generic as cg

    b, c, n, h, w, kh, kw, st, pd = case
    g = torch.Generator().manual_seed(7)
    x0, w0, b0 = torch.randn(b, c, h, w, generator=g), torch.randn(n, c, kh, kw, generator=g) / (c * kh * kw) ** 0.5, \
        torch.randn(n, generator=g)

    def run(dev, dt):
        x, wt, bias = [t.to(dev, dt).requires_grad_(True) for t in (x0, w0, b0)]
        conv = cg.conv2d_generic if dev == DEV else (lambda a, ww, bb, s, p: F.conv2d(a, ww, bb, stride=s, padding=p))
        y = conv(x, wt, bias, st, pd)
        proj = torch.randn(y.shape, generator=torch.Generator().manual_seed(8)).to(dev, dt)
        gx, gw, gb = torch.autograd.grad((y * proj).sum(), [x, wt, bias], create_graph=True)
        # second order: gradient of |gx|^2 + |gw|^2 w.r.t. the weights and the projection-free input
        (hw, hx) = torch.autograd.grad((gx * gx).sum() + (gw * gw).sum(), [wt, x])
        return [t.detach().double().cpu() for t in (y, gx, gw, gb, hw, hx)]

    got = run(DEV, torch.float32)
    want = run("cpu", torch.float64)
    # scale of each result: the same expression on absolute values bounds sum|a*b|
    ya = F.conv2d(x0.double().abs(), w0.double().abs(), None, stride=st, padding=pd)
    bound = float(ya.max())
    for name, a, r in zip(("y", "gx", "gw", "gb", "hw", "hx"), got, want):
        scale = max(float(r.abs().max()), bound if name == "y" else 0.0)
        err = float((a - r).abs().max())
        assert err <= (2e-6 if name in ("y", "gx", "gw") else 2e-5) * scale * (1.0 if name == "y" else 8.0), (name, err, scale)
    # transposed convolution as a forward operator (the up-sampling ModulatedConv2d)
    wt_t = torch.randn(c, n, kh, kw, generator=g) / (c * kh * kw) ** 0.5
    yt = cg.conv_transpose2d_generic(x0.to(DEV), wt_t.to(DEV), stride=st, padding=pd)
    rt = F.conv_transpose2d(x0.double(), wt_t.double(), stride=st, padding=pd)
    assert yt.shape == rt.shape
    assert float((yt.double().cpu() - rt).abs().max()) <= 2e-6 * float(
        F.conv_transpose2d(x0.double().abs(), wt_t.double().abs(), stride=st, padding=pd).max())


def test_generic_geometry_layers_vs_reference(golden):
    """layers.ModulatedConv2d with kernel_size 5 / 7 (plain, up-sampling incl. the cropping blur, down-sampling, no
    demodulation) and EqualConv2d 5x5 p2 / 4x4 s2 p1 / 3x3 p0 / 2x2 s3 on device tensors — under SR_STRICT_NATIVE, i.e.
    without MIOpen / rocBLAS — against the reference's layers (tests/golden/conv_generic.npz): output and every gradient
    at 2e-6 of the tensor's scale (5e-6 for the parameter gradients of the modulated layers, which pass through the
    demodulation's cancellation)."""
    from stylerenderer_amd import layers, synth

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))  # noqa: E731
    gold = golden("conv_generic")
    for tag, kw in [("m5", dict(kernel_size=5)), ("m5up", dict(kernel_size=5, upsample=True)),
                    ("m5down", dict(kernel_size=5, downsample=True)), ("m7up", dict(kernel_size=7, upsample=True)),
                    ("m5nodemod", dict(kernel_size=5, demodulate=False))]:
        m = layers.ModulatedConv2d(in_channel=8, out_channel=12, style_dim=16, **kw)
        synth.fill_state_dict(m.state_dict(), salt=35)
        m = m.to(DEV)
        x, s = T(synth.det_normal((2, 8, 10, 10), 36)).requires_grad_(), T(synth.det_normal((2, 16), 37)).requires_grad_()
        y = m(x, s)
        assert rel(y.detach().cpu().numpy(), gold[tag + "_y"]) < 2e-6, tag
        gy = T(synth.det_normal(tuple(y.shape), 38))
        grads = torch.autograd.grad(y, [x, s, m.weight, m.modulation.weight, m.modulation.bias], gy)
        for a, k in zip(grads, ("gx", "gs", "gw", "gmw", "gmb")):
            assert rel(a.cpu().numpy(), gold[tag + "_" + k]) < 5e-6, (tag, k)
    for tag, (k, st, pd) in [("e5", (5, 1, 2)), ("e4s2", (4, 2, 1)), ("e3p0", (3, 1, 0)), ("e2s3", (2, 3, 0))]:
        m = layers.EqualConv2d(6, 10, k, stride=st, padding=pd)
        synth.fill_state_dict(m.state_dict(), salt=39)
        m = m.to(DEV)
        x = T(synth.det_normal((3, 6, 11, 13), 40)).requires_grad_()
        y = m(x)
        assert rel(y.detach().cpu().numpy(), gold[tag + "_y"]) < 2e-6, tag
        grads = torch.autograd.grad(y, [x, m.weight, m.bias], T(synth.det_normal(tuple(y.shape), 41)))
        for a, kk in zip(grads, ("gx", "gw", "gb")):
            assert rel(a.cpu().numpy(), gold[tag + "_" + kk]) < 2e-6, (tag, kk)


@pytest.mark.parametrize("b,c,n,h,w", [(2, 32, 128, 16, 32), (3, 48, 256, 8, 16), (1, 16, 128, 32, 64)])
@pytest.mark.parametrize("scaled", [True, False])
def test_conv1x1_gemm_vs_float64_and_window_kernel(b, c, n, h, w, scaled, monkeypatch):
    """csrc/conv1x1_gemm.hip (the discriminator's skip convolutions as a plain GEMM: reference model.py:296-336) against
    float64 and against the 1x1 instantiation of k_conv_mfma it replaces."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    g = torch.Generator().manual_seed(b * 31 + c + n + h)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(n, c, 1, 1, generator=g) / np.sqrt(c)
    isc = torch.randn(b, c, generator=g) if scaled else None
    osc = (torch.rand(b, n, generator=g) + 0.5) if scaled else None
    bias = torch.randn(n, generator=g) if scaled else None
    want = ref_conv(x, wgt, isc, osc, bias, 1, 0, False)
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    args = [dev(x), dev(to_taps(wgt, False)), dev(isc), dev(osc), dev(bias)]
    monkeypatch.setenv("SR_CONV1X1_GEMM", "force")
    got = conv2d_mfma(*args, 1, 1, 0, False)
    mag = F.conv2d(x.abs().double() * (isc.abs().double()[:, :, None, None] if scaled else 1.0), wgt.abs().double())
    if scaled:
        mag = mag * osc.abs().double()[:, :, None, None] + bias.abs().double()[None, :, None, None]
    assert float(((got.cpu().double() - want).abs() / (mag + 1e-30)).max()) < 2e-6
    assert torch.equal(got, conv2d_mfma(*args, 1, 1, 0, False))
    monkeypatch.setenv("SR_CONV1X1_GEMM", "0")
    old = conv2d_mfma(*args, 1, 1, 0, False)
    assert float(((old - got).abs().cpu().double() / (mag + 1e-30)).max()) < 2e-6


@pytest.mark.parametrize("b,c,n,h,w", [(4, 64, 128, 4, 4), (2, 128, 256, 8, 8), (3, 96, 128, 16, 16), (1, 32, 128, 32, 32),
                                       (2, 48, 128, 5, 9)])
@pytest.mark.parametrize("scaled", [True, False])
def test_convt_taps_gemm_vs_float64_and_patch_kernel(b, c, n, h, w, scaled, monkeypatch):
    """Tap-split transposed convolution with the flattened (sample, grid point) pixel dimension (csrc/conv1x1_gemm.hip
    k_convt_taps_gemm) against float64 and against the patch form it replaces (same K slices, same reduction)."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    g = torch.Generator().manual_seed(b * 131 + c + n + h * 7 + w)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(c, n, 3, 3, generator=g) / (3 * c ** 0.5)
    isc = torch.randn(b, c, generator=g) if scaled else None
    osc = torch.randn(b, n, generator=g) if scaled else None
    bias = torch.randn(n, generator=g) if scaled else None
    want = ref_conv(x, wgt, isc, osc, bias, 2, 0, True)
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    args = [dev(x), dev(to_taps(wgt, True)), dev(isc), dev(osc), dev(bias)]
    monkeypatch.setenv("SR_CONVT_TAPS", "1")
    got = conv2d_mfma(*args, 3, 2, 0, True)
    mag = F.conv_transpose2d(x.abs().double() * (isc.abs().double()[:, :, None, None] if scaled else 1.0), wgt.abs().double(),
                             stride=2)
    if scaled:
        mag = mag * osc.abs().double()[:, :, None, None] + bias.abs().double()[None, :, None, None]
    assert float(((got.cpu().double() - want).abs() / (mag + 1e-30)).max()) < 2e-6
    assert torch.equal(got, conv2d_mfma(*args, 3, 2, 0, True))
    monkeypatch.setenv("SR_CONVT_TAPS_GEMM", "0")
    old = conv2d_mfma(*args, 3, 2, 0, True)
    assert float(((old - got).abs().cpu().double() / (mag + 1e-30)).max()) < 2e-6


def test_conv1x1_add_fused_equals_the_two_operators_incl_gradients(monkeypatch):
    """op.conv.conv1x1_add (ResBlock: skip convolution + the other branch in one store, reference model.py ResBlock.forward):
    output and all first-order gradients equal the separate convolution + addition bit for bit; a second-order probe
    (R1-style: gradient of the input gradient's square) agrees too."""
    from stylerenderer_amd.op import conv as C

    monkeypatch.setenv("SR_CONV1X1_GEMM", "force")
    g = torch.Generator().manual_seed(3)
    b, c, n, h, w = 2, 32, 128, 16, 16
    wt0 = (torch.randn(1, c, n, generator=g) / c ** 0.5).to(DEV)
    osc = torch.full((b, n), 0.70710678, device=DEV)
    x0, a0 = torch.randn(b, c, h, w, generator=g).to(DEV), torch.randn(b, n, h, w, generator=g).to(DEV)
    gy = torch.randn(b, n, h, w, generator=g).to(DEV)

    def run(fused):
        x, wt, a = x0.clone().requires_grad_(True), wt0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
        out = C.conv1x1_add(x, wt, osc, a) if fused else C.ConvFn.apply(x, wt, None, osc, None, "c1") + a
        gx, gw, ga = torch.autograd.grad(out, (x, wt, a), gy, create_graph=True)
        (g2,) = torch.autograd.grad((gx * gx).sum(), wt)
        return out.detach(), gx.detach(), gw.detach(), ga.detach(), g2

    assert isinstance(C.conv1x1_add(x0.clone().requires_grad_(True), wt0, osc, a0).grad_fn, C.Conv1x1AddFn.apply.__self__._backward_cls)
    for got, want in zip(run(True), run(False)):
        assert torch.equal(got, want)
