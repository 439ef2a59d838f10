"""GPU: the full networks on the HIP path against the golden vectors generated from the reference.
Tolerance (north_star: "within a stated fp32 tolerance for generator activations"): the modulated
convolution applies style/demodulation to the operands instead of building per-sample weights, and
the MFMA accumulates in a different order than oneDNN, so activations agree to fp32 round-off:
   max |a - b| <= 2e-4 * max |b|      (images, latents, normal maps; measured 2e-6 at 256x256)
   gradients: 256 evenly spaced entries of EVERY gradient tensor against the reference's, relative to the tensor's
   scale — first order 5e-6 (measured 1.2e-6), path-length double backward 4e-5 (measured 1.5e-6 here, 9e-6 on the CPU path: scalar noise strengths), R1 double
   backward 5e-5 (measured 1.1e-5) — and FULL tensors against this repo's own CPU formulation."""
import numpy as np
import pytest
import torch

from stylerenderer_amd import model, synth
from test_model_cpu import check_grad_digest, noise_list
from util import check_grad_samples, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-4


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def g8():
    g = model.Generator(8, 64, 2)
    synth.fill_state_dict(g.state_dict(), salt=41)
    return g.to(DEV)


def dev_noise(g, key):
    return [n.to(DEV) for n in noise_list(g, key)]


def test_generator_s8_forward_variants(golden, g8):
    gold = golden("generator_s8")
    z = T(synth.det_normal((2, 64), 42))
    noise = dev_noise(g8, 4300)
    img, lat = g8([z], return_latents=True, noise=noise)
    assert rel_err(lat.detach().cpu().numpy(), gold["latent"]) < 1e-5
    assert rel_err(img.detach().cpu().numpy(), gold["image"]) < TOL
    img2, _ = g8([z, T(synth.det_normal((2, 64), 44))], inject_index=1, noise=noise)
    assert rel_err(img2.detach().cpu().numpy(), gold["image_mix"]) < TOL
    img3, _ = g8([z], randomize_noise=False)
    assert rel_err(img3.detach().cpu().numpy(), gold["image_bufnoise"]) < TOL
    img4, _ = g8([z], truncation=0.7, truncation_latent=T(gold["trunc_latent"]), noise=noise)
    assert rel_err(img4.detach().cpu().numpy(), gold["image_trunc"]) < TOL


def test_generator_s8_gradients(golden, g8):
    gold = golden("generator_s8")
    z = T(synth.det_normal((2, 64), 42))
    img, _ = g8([z], noise=dev_noise(g8, 4300))
    proj = T(synth.det_normal(tuple(img.shape), 46))
    params = dict(g8.named_parameters())
    grads = torch.autograd.grad((img * proj).sum(), list(params.values()), allow_unused=True)
    got = {n: g for n, g in zip(params, grads) if g is not None}
    assert sorted(n for n, g in zip(params, grads) if g is None) == list(gold["unused"])
    check_grad_digest(got, gold["grad_names"], gold["grad_norms"], gold["grad_heads"], 1e-3)
    check_grad_samples(got, gold["grad_names"], gold["grad_samples"], gold["grad_sample_offsets"], 5e-6)


def test_path_length_regulariser_double_backward(golden, g8):
    gold = golden("generator_s8")
    z = T(synth.det_normal((2, 64), 42))
    img, lat = g8([z], return_latents=True, noise=dev_noise(g8, 4300))
    pl_noise = T(synth.det_normal(tuple(img.shape), 47)) / np.sqrt(img.shape[2] * img.shape[3])
    (gl,) = torch.autograd.grad((img * pl_noise).sum(), lat, create_graph=True)
    flat = gl.reshape(gl.shape[0], -1)
    lengths = torch.sqrt((flat * flat).sum(1))
    assert rel_err(lengths.detach().cpu().numpy(), gold["pl_lengths"]) < 1e-3
    penalty = (lengths - 0.01 * lengths.mean()).pow(2).mean()
    assert abs(penalty.item() - float(gold["pl_penalty"])) < 2e-3 * float(gold["pl_penalty"])
    g8.zero_grad()
    penalty.backward()
    got = {n: p.grad for n, p in g8.named_parameters() if p.grad is not None}
    check_grad_digest(got, gold["pl_grad_names"], gold["pl_grad_norms"], gold["pl_grad_heads"], 3e-3)
    check_grad_samples(got, gold["pl_grad_names"], gold["pl_grad_samples"], gold["pl_grad_sample_offsets"], 4e-5)
    g8.zero_grad()


def test_generator_s64_vs_reference_image(golden):
    gold = golden("generator_s64")
    g = model.Generator(64, 512, 8)
    synth.fill_state_dict(g.state_dict(), salt=41)
    g = g.to(DEV)
    with torch.no_grad():
        img, lat = g([T(synth.det_normal((1, 512), 42))], return_latents=True, noise=dev_noise(g, 4300))
    assert rel_err(img.cpu().numpy(), gold["image"]) < TOL


def test_generator_gpu_equals_own_cpu_path_at_64():
    """Same module, CPU tensors (reference formulation) vs device tensors (HIP kernels), B=4."""
    g = model.Generator(64, 512, 8)
    synth.fill_state_dict(g.state_dict(), salt=3)
    z = torch.from_numpy(synth.det_normal((4, 512), 9))
    noise = noise_list(g, 77)
    with torch.no_grad():
        a, _ = g([z], noise=noise)
    g = g.to(DEV)
    with torch.no_grad():
        b, _ = g([z.to(DEV)], noise=[n.to(DEV) for n in noise])
    assert rel_err(b.cpu().numpy(), a.numpy()) < TOL


@pytest.fixture(params=["exact", "split_bf16"])
def conv_arith(request, monkeypatch):
    """The reference fixtures of the big networks run twice: on the exact-fp32 MFMA kernels (the default), and with the
    opt-in split-bf16 kernels (SR_CONV_SPLIT_BF16=1, read per call: stride-2 / transposed 3x3 convolutions, stride-2 and
    1x1 weight gradients) — at the SAME bars: the flag is a model-level claim only if the whole network
    holds them."""
    monkeypatch.setenv("SR_CONV_SPLIT_BF16", "1" if request.param == "split_bf16" else "0")
    return request.param


@pytest.mark.parametrize("size", [16, 64, 256])
def test_generator_with_map_vs_reference_incl_gradients(golden, size, conv_arith):
    """M7 on the HIP path against the reference (oracle/make_golden._gwm_case): image, normal maps, gradients of
    every parameter (sampled) and of the mesh (full grad_v / grad_nrm through sr_rasterize_grad and the map heads),
    and one g_path_regularize(img, [latents] + norm_maps) evaluation (reference train.py:340-347) — lengths, penalty,
    running mean, double-backward parameter gradients (k_nba_aff_bwd2, k_rowdot_bwd) and the mesh gradient of the
    penalty.  64^2: the 32^2 / 64^2 layers run the Winograd and LDS-tiled FIR variants."""
    from util import run_generator_with_map_case

    meas = run_generator_with_map_case(golden("generator_map_s%d" % size), size, DEV,
                                       1e-5, 2e-5, 4e-5, 2e-5, 2e-5)   # measured: 1.5e-6, 5e-6, 5e-6, 1e-6, 1e-6
    print(size, conv_arith, meas)
    assert meas.get("forced", 0) <= 64, meas


def test_discriminator_s16(golden):
    gold = golden("discriminator_s16")
    d = model.Discriminator(16)
    synth.fill_state_dict(d.state_dict(), salt=61)
    d = d.to(DEV)
    x = T(gold["x"]).requires_grad_()
    y = d(x)
    assert rel_err(y.detach().cpu().numpy(), gold["y"]) < TOL
    (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    assert rel_err(gx.detach().cpu().numpy(), gold["gx"]) < 5e-4
    r1 = (gx * gx).reshape(4, -1).sum(1).mean()
    assert abs(r1.item() - float(gold["r1"])) < 1e-3 * float(gold["r1"])
    d.zero_grad()
    r1.backward()
    got = {n: p.grad for n, p in d.named_parameters() if p.grad is not None}
    check_grad_digest(got, gold["r1_grad_names"], gold["r1_grad_norms"], gold["r1_grad_heads"], 3e-3)
    check_grad_samples(got, gold["r1_grad_names"], gold["r1_grad_samples"], gold["r1_grad_sample_offsets"], 5e-5)


def test_generator_256_vs_reference_image_and_gradients(golden, conv_arith):
    """The network bench.py times (Generator(256, 512, 8)) against the reference (tests/golden/generator_s256.npz):
    image of one latent (1e-5), and the gradients of <img, proj> w.r.t. every parameter (256 samples per tensor) and
    the W+ latent (full tensor) —
      * with linear activations (every FusedLeakyReLU at negative_slope = 1, in the reference too): 2e-5 of each
        tensor's scale.  This is the pin of the 128^2 / 256^2 Winograd forward / data-gradient / weight-gradient,
        k_wgrad_s2_dma, k_convt_fused backward, fused up-sampling and ToRGB variants: exact adjoints at round-off;
      * at the real slope 0.2, KINK-AWARE: a few dozen of the 3.3e7 pre-activations sit within 1e-7 of zero and take
        different sides under any two fp32 implementations.  The fixture records, per layer, the outputs within
        5e-5 of the layer's scale of the kink with the reference's side, and the per-channel positive counts
        (oracle/make_golden.KinkRecorder); util.KinkForcer puts the disagreeing ones on the reference's side, asserts
        that no disagreement exists outside that band, and the mask-dependent backward (k_nba_bwd with its rebuilt
        y0, the fused tails) is then compared at the SAME 2e-5 of each tensor's scale as the linear pass (measured 5.9e-6
        with 21 of 3.3e7 elements forced; the bar was 2e-2 while the kink was not controlled, measured 6e-3)."""
    from stylerenderer_amd.op import FusedLeakyReLU
    from util import KinkForcer

    gold = golden("generator_s256")
    g = model.Generator(256, 512, 8)
    assert len(g.state_dict()) == int(gold["n_keys"])
    synth.fill_state_dict(g.state_dict(), salt=41)
    g = g.to(DEV)
    proj = None
    for prefix, slope in (("grad", 0.2), ("lin_grad", 1.0)):
        for m in g.modules():
            if isinstance(m, FusedLeakyReLU):
                m.negative_slope = slope
        forcer = KinkForcer(g, gold) if slope != 1.0 else None
        img, lat = g([T(synth.det_normal((1, 512), 42))], return_latents=True, noise=dev_noise(g, 4300))
        if forcer is not None:
            forcer.close()
            assert len(forcer.stats) == len(gold["kink_keys"]) == g.num_layers + 8
            assert not forcer.unexplained(), forcer.unexplained()      # sign patterns identical after forcing
            assert forcer.disagreements() <= 64, forcer.stats          # a few dozen of 3.3e7
        if proj is None:
            proj = T(synth.det_normal(tuple(img.shape), 46))
            assert rel_err(lat[0, 0].detach().cpu().numpy(), gold["latent_row"]) < 1e-5
            assert rel_err(img.detach().cpu().numpy(), gold["image"]) < 1e-5            # measured 2.4e-6
        else:
            assert rel_err(img.detach().cpu().numpy()[:, :, ::4, ::4], gold["lin_image"]) < 1e-5
        params = dict(g.named_parameters())
        grads = torch.autograd.grad((img * proj).sum(), list(params.values()) + [lat], allow_unused=True)
        got = {n: x for n, x in zip(params, grads[:-1]) if x is not None}
        names, vals, offs = gold[prefix + "_names"], gold[prefix + "_samples"], gold[prefix + "_sample_offsets"]
        e_lat = rel_err(grads[-1].cpu().numpy(), gold[prefix + "_latent"])
        bar = 2e-5
        worst = check_grad_samples(got, names, vals, offs, bar)       # (one-element tensors: vs the largest of them)
        assert e_lat < bar, e_lat
        print("256^2 gradients, slope %.1f: worst sampled %.2e, latent %.2e%s" % (
            slope, worst, e_lat, "" if forcer is None else ", kink elements forced %d" % forcer.disagreements()))


def test_generator_256_batch16_vs_reference(golden, conv_arith):
    """BASELINE config[1] at its real batch against the reference (tests/golden/generator_s256_b16.npz, written by
    oracle/make_golden.gold_generator_256_b16): Generator(256, 512, 8) on 16 latents through the mapping network with
    per-sample noise maps — every 8th pixel of all 16 images at the real slope (1e-5), and, with linear activations
    (no kink enters; the mapping network keeps its slope, as in the B = 1 test), 256 samples of every parameter
    gradient — each a sum over the 16 samples, i.e. the batch reduction of every weight-gradient kernel at exactly the
    launch shapes bench.py times — and the full gradient w.r.t. the 16 mapped latents, at the 2e-5 of the B = 1 pin."""
    from stylerenderer_amd.op import FusedLeakyReLU

    gold = golden("generator_s256_b16")
    b = 16
    g = model.Generator(256, 512, 8)
    synth.fill_state_dict(g.state_dict(), salt=41)
    g = g.to(DEV)
    z = T(synth.det_normal((b, 512), 47))
    noise = [T(synth.det_normal((b, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)), 4400 + i)) for i in range(g.num_layers)]
    with torch.no_grad():
        img, _ = g([z], noise=noise)
        w = g.style(z)
    e_img = rel_err(img.cpu().numpy()[:, :, ::8, ::8], gold["image_s8"])
    e_w = rel_err(w.cpu().numpy(), gold["w"])
    assert e_img < 1e-5 and e_w < 1e-5, (e_img, e_w)
    del img
    for m in g.modules():
        if isinstance(m, FusedLeakyReLU):
            m.negative_slope = 1.0
    w = T(gold["w"]).requires_grad_(True)
    img, _ = g([w], input_is_latent=True, noise=noise)
    e_lin = rel_err(img.detach().cpu().numpy()[:, :, ::16, ::16], gold["lin_image_s16"])
    assert e_lin < 1e-5, e_lin
    proj = T(synth.det_normal((b, 3, 32, 32), 48)).repeat_interleave(8, 2).repeat_interleave(8, 3)
    params = dict(g.named_parameters())
    grads = torch.autograd.grad((img * proj).sum(), list(params.values()) + [w], allow_unused=True)
    got = {n: x for n, x in zip(params, grads[:-1]) if x is not None}
    worst = check_grad_samples(got, gold["lin_grad_names"], gold["lin_grad_samples"], gold["lin_grad_sample_offsets"], 2e-5)
    e_gw = rel_err(grads[-1].cpu().numpy(), gold["lin_grad_w"])
    assert e_gw < 2e-5, e_gw
    print("256^2 B=16: image %.2e, w %.2e, linear image %.2e, worst sampled gradient %.2e, latent gradient %.2e" % (
        e_img, e_w, e_lin, worst, e_gw))


@pytest.mark.parametrize("size", [64, 128, 256])
def test_discriminator_big_vs_reference(golden, size, conv_arith):
    """N1 at sizes where the big kernels run (reference model.py:296-336, layers.py:341-391): D(64) / D(128), batch 4,
    against tests/golden/discriminator_s<size>.npz — logits, first-order gradients of every parameter (256 samples
    each) and of the input, and ONE R1 evaluation as the step weights it (train.py:110-114, 281-289) with its
    double-backward parameter gradients.  In the chain: shared-weight Winograd 3x3 convolutions with bias + LeakyReLU
    in the store (ConvNBAFn) and their recorded backward, c3s2 after Blur pad (2, 2), k_fir4_resample<1,2> + c1 skip
    (SkipDown fork), minibatch-stddev.  Kink-aware like the 256^2 generator test (util.KinkForcer).  256: the per-GPU
    batch of BASELINE config[2] at its real size — the 3 -> 128 1x1 at 256^2, the first 128-channel Winograd pair and
    the 257^2 blur exist only there."""
    from util import run_discriminator_case

    meas = run_discriminator_case(golden("discriminator_s%d" % size), size, DEV, 2e-5, 5e-5, 1e-4)
    print("D(%d) %s:" % (size, conv_arith), meas)
    assert meas["forced"] <= (64 if size < 256 else 192), meas


def _activation_signs(net, store):
    """Forward hooks recording the sign pattern of every leaky-ReLU output (StyledConv tails, mapping network)."""
    from stylerenderer_amd import layers

    hooks = []
    for name, m in net.named_modules():
        if isinstance(m, model.StyledConv) or (isinstance(m, layers.EqualLinear) and m.activation):
            hooks.append(m.register_forward_hook(
                lambda mod, inp, out, name=name: store.__setitem__(name, (out.detach() > 0).cpu())))
    return hooks


@pytest.mark.parametrize("size,batch,slope,tol", [(16, 3, 1.0, 2e-5), (64, 2, 1.0, 2e-5), (16, 3, 0.2, 2e-5)])
def test_full_gradient_tensors_gpu_vs_own_cpu_path(size, batch, slope, tol):
    """Every gradient tensor IN FULL: device tensors (HIP kernels, operand-scaled shared weights) against the same
    module on CPU tensors (the reference's grouped-convolution formulation, itself pinned to the reference's
    gradients in tests/test_model_cpu.py), at fp32 round-off: 2e-5 of each tensor's scale.
    slope = 1.0 makes every LeakyReLU linear: all the operators in between are compared as exact adjoints.
    With the real slope 0.2 a pre-activation of |value| ~ 1e-7 can land on different sides of the kink in the two fp32
    forward passes; that one element's derivative is then 1 on one path and 0.2 on the other and every upstream
    gradient moves by ~1e-3 — a property of LeakyReLU, not an error of either path.  The sign pattern of EVERY
    activation output is therefore recorded on both paths (forward hooks) and the comparison runs on the first latent
    batch whose patterns agree element for element (at most a few of the ~4e5 pre-activations per batch ever differ),
    at the SAME 2e-5 bar as the linear case — a 1 % systematic error cannot hide behind a widened tolerance."""
    from stylerenderer_amd.op import FusedLeakyReLU

    g = model.Generator(size, 64, 2)
    synth.fill_state_dict(g.state_dict(), salt=5)
    for m in g.modules():
        if isinstance(m, FusedLeakyReLU):
            m.negative_slope = slope
    noise = noise_list(g, 70)
    proj = torch.from_numpy(synth.det_normal((batch, 3, size, size), 7))
    g_dev = model.Generator(size, 64, 2)
    g_dev.load_state_dict(g.state_dict())
    for m in g_dev.modules():
        if isinstance(m, FusedLeakyReLU):
            m.negative_slope = slope
    g_dev = g_dev.to(DEV)

    def grads(net, dev, z):
        signs = {}
        hooks = _activation_signs(net, signs)
        img, _ = net([z.to(dev)], noise=[n.to(dev) for n in noise])
        for h in hooks:
            h.remove()
        params = dict(net.named_parameters())
        out = torch.autograd.grad((img * proj.to(dev)).sum(), list(params.values()), allow_unused=True)
        return {n: o for n, o in zip(params, out) if o is not None}, signs

    flipped = []
    for key in range(6, 14):
        z = torch.from_numpy(synth.det_normal((batch, 64), key))
        want, s_cpu = grads(g, "cpu", z)
        got, s_dev = grads(g_dev, DEV, z)
        assert sorted(s_cpu) == sorted(s_dev) and len(s_cpu) >= g.num_layers + 1
        n_flip = sum(int((s_cpu[k] != s_dev[k]).sum()) for k in s_cpu)
        flipped.append(n_flip)
        if n_flip == 0 or slope == 1.0:
            break
    assert flipped[-1] == 0 or slope == 1.0, "no latent batch without a kink flip: %s" % flipped
    assert max(flipped) <= 8, flipped                 # a handful of ~4e5 pre-activations at most
    assert sorted(got) == sorted(want)
    for n in want:
        scale = float(want[n].abs().max())
        # scalar noise strengths are sums of ~4e5 signed terms: 5x the bar (measured 2.1e-5; all others <= 3.6e-6)
        t = tol * (5 if want[n].numel() == 1 else 1)
        err = float((got[n].cpu() - want[n]).abs().max())
        assert err <= t * scale + 1e-9, "%s: |err| %.3e = %.3e of the tensor's scale %.3e (bar %.1e); flips %s" % (
            n, err, err / max(scale, 1e-30), scale, t, flipped)


@pytest.mark.parametrize("tag,kw", [("plain", dict(in_channel=8, out_channel=6, kernel_size=3, style_dim=16)),
                                    ("up", dict(in_channel=8, out_channel=6, kernel_size=3, style_dim=16, upsample=True)),
                                    ("rgb", dict(in_channel=8, out_channel=3, kernel_size=1, style_dim=16,
                                                 demodulate=False))])
def test_modulated_conv_layer_vs_reference_gradients(golden, tag, kw, monkeypatch):
    """layers.ModulatedConv2d on the HIP path against the reference layer's output and ALL its gradients (input,
    style, weight, modulation weight / bias), full tensors (tests/golden/modconv.npz).  Measured <= 3.2e-7.
    (SURVEY 8(c)'s fixture has 6 output channels: the convolution kernels take it, the [2, 8] x [8, 6] demodulation
    product is outside sr_demod_fwd's 4-column granularity and runs on the library — the one shape of the suite that
    needs SR_STRICT_NATIVE off; no layer of G / D has a channel count that is not a multiple of 4.)"""
    from stylerenderer_amd import layers

    monkeypatch.setenv("SR_STRICT_NATIVE", "0")

    gold = golden("modconv")
    m = layers.ModulatedConv2d(**kw)
    synth.fill_state_dict(m.state_dict(), salt=31)
    m = m.to(DEV)
    x, s = T(gold[tag + "_x"]).requires_grad_(), T(gold[tag + "_s"]).requires_grad_()
    y = m(x, s)
    assert rel_err(y.detach().cpu().numpy(), gold[tag + "_y"]) < 2e-6
    grads = torch.autograd.grad(y, [x, s, m.weight, m.modulation.weight, m.modulation.bias], T(gold[tag + "_gy"]))
    for a, k in zip(grads, ("gx", "gs", "gw", "gmw", "gmb")):
        assert rel_err(a.cpu().numpy(), gold[tag + "_" + k]) < 2e-6, k


def test_generator_256_full_size_properties():
    """BASELINE config 1 size (256x256, B=16): fwd+bwd runs on the HIP path; per-sample
    independence (sample 5 of the batch == the same latent alone) and finite gradients."""
    g = model.Generator(256, 512, 8).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(0)
    z = torch.randn(16, 512, device=DEV, generator=gen)
    noise = [n.to(DEV) for n in noise_list(g, 900)]
    img, _ = g([z], noise=noise)
    assert img.shape == (16, 3, 256, 256)
    img.sum().backward()
    for n, p in g.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
    with torch.no_grad():
        one, _ = g([z[5:6]], noise=noise)
    assert rel_err(one.cpu().numpy(), img[5:6].detach().cpu().numpy()) < TOL


def test_resblock_fork_node_equals_two_consumers(monkeypatch):
    """ResBlock with the fused fork (op.upfirdn2d.SkipDown: the two input gradients are added inside the up-sampling FIR
    kernel of the skip branch) against the plain two-consumer form: same output, same first- and second-order gradients
    (R1) bit for bit — the fused kernel adds the same two values."""
    from stylerenderer_amd.layers import ResBlock

    torch.manual_seed(3)
    blk = ResBlock(8, 16).to("cuda")
    x0 = torch.randn(2, 8, 32, 32, device="cuda")
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SR_SKIP_FUSED", mode)
        x = x0.clone().requires_grad_()
        y = blk(x)
        (gx,) = torch.autograd.grad(y.square().sum(), x, create_graph=True)
        for p in blk.parameters():
            p.grad = None
        gx.square().sum().backward()
        res[mode] = (y.detach(), gx.detach(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()])
    assert torch.equal(res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1])
    assert torch.equal(res["1"][2], res["0"][2])
    for a, b in zip(res["1"][3], res["0"][3]):
        assert torch.equal(a, b)


def test_torgb_fork_equals_two_consumers_bit_for_bit(monkeypatch):
    """op.smallconv.SmallConvFork (the feature map goes on to the next layer through the ToRGB node, whose backward adds
    the map's two gradients inside the data-gradient kernel) against ToRGB as a second consumer with autograd's own
    addition (reference model.py:206-219): image and every parameter gradient identical."""
    import torch

    from stylerenderer_amd.model import Generator

    dev = torch.device("cuda")
    torch.manual_seed(9)
    g = Generator(64, 64, 2).to(dev)
    z = torch.randn(3, 64, device=dev)
    probe = torch.randn(3, 3, 64, 64, device=dev)

    def run(fork):
        monkeypatch.setenv("SR_TORGB_FORK", "1" if fork else "0")
        for p in g.parameters():
            p.grad = None
        img, _ = g([z], randomize_noise=False)
        (img * probe).sum().backward()
        return img.detach().clone(), {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None}

    img_a, ga = run(True)
    img_b, gb = run(False)
    assert torch.equal(img_a, img_b)
    assert set(ga) == set(gb) and len(ga) > 30
    assert not [k for k in ga if not torch.equal(ga[k], gb[k])]
