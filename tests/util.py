"""Helpers shared by the tests."""
import numpy as np


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def max_ulp(a, b):
    """Largest distance in float32 units-in-the-last-place between two arrays."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)

    def key(x):
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, np.int64(-(2 ** 31)) - i, i)

    return int(np.abs(key(a) - key(b)).max()) if a.size else 0


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / den


def check_grad_samples(named_grads, names, values, offsets, tol, scalar_factor=1.0, truth=None):
    """Sampled entries of every gradient tensor (fixtures written by oracle/make_golden.grad_samples): the error of
    each tensor's samples relative to that tensor's largest sampled magnitude.  Returns the worst ratio.
    One-element tensors (noise strengths) are measured against the LARGEST of them: each is one sum of up to 2e6
    signed terms of similar size, and the ones that cancel to a few percent of their terms (|value| 0.1 beside 20-1000
    for their siblings) carry the summation order's absolute error — also between two CPUs running the same code — at
    a relative size that says nothing about the kernel.  `scalar_factor` widens their bar further.
    `truth` (float64 samples of the same entries: the reference run in float64 under its float32 sign pattern,
    oracle/make_golden.KinkPinned): the error is then measured against the EXACT value, and a tensor's bar is the larger
    of `tol` and twice the distance of the reference's own float32 result from it — a float32 implementation is not asked
    to reproduce the reference's rounding noise where that noise exceeds the bar (second-order bias gradients: sums of
    ~1e6 cancelling terms, 2e-4 .. 7e-4 of their scale in the reference itself)."""
    from stylerenderer_amd import synth

    assert sorted(named_grads) == list(names)
    worst = 0.0
    scalars = [float(np.abs(values[offsets[i]:offsets[i + 1]]).max()) for i, n in enumerate(names)
               if named_grads[n].numel() == 1]
    scalar_scale = max(scalars) if scalars else 0.0
    for i, n in enumerate(names):
        ref = np.asarray(values[offsets[i]:offsets[i + 1]], np.float64)
        want = ref if truth is None else np.asarray(truth[offsets[i]:offsets[i + 1]], np.float64)
        g = named_grads[n].detach().reshape(-1).cpu().numpy().astype(np.float64)
        got = g[synth.sample_index(g.size, 256)]
        assert got.shape == want.shape, n
        scale = max(float(np.abs(want).max()), scalar_scale if g.size == 1 else 0.0, 1e-12)
        err = float(np.abs(got - want).max()) / scale
        bar = tol * (scalar_factor if g.size == 1 else 1.0)
        ref_err = float(np.abs(ref - want).max()) / scale
        bar = max(bar, 2.0 * ref_err)
        assert err <= bar, "%s: sampled-gradient error %.3e of the tensor's scale (bar %.1e; reference's own float32 " \
                           "error %.1e)" % (n, err, bar, ref_err)
        worst = max(worst, err / bar * tol)
    return worst


GWM_CASES = {
    # size: (style_dim, n_mlp, batch, zkey, nkey, salt)
    16: (64, 2, 2, 52, 5300, 51),
    64: (64, 2, 1, 61, 5700, 53),
    256: (512, 8, 1, 71, 6100, 57),        # BASELINE config[2] / config[4]'s network on the face-sized mesh
}


def run_generator_with_map_case(gold, tag_size, dev, tol_img, tol_g1, tol_g2, tol_mesh1, tol_mesh2):
    """GeneratorWithMap against tests/golden/generator_map_s<size>.npz (written by oracle/make_golden._gwm_case /
    gold_generator_with_map_256 from the reference's model.GeneratorWithMap + train.g_path_regularize): image, normal
    maps, first-order gradients of <img, proj> w.r.t. every parameter (sampled) and the mesh (full tensors), and one
    path-length regulariser evaluation over [latents] + normal maps with its double-backward gradients.  The 256^2
    fixture rebuilds its mesh from integers (synth.face_sized_mesh, digest-checked), holds sub-sampled image / maps,
    carries LeakyReLU kink records (KinkForcer) and redraws the regulariser's probe from torch's CPU generator.
    Returns the measured errors."""
    import torch

    from stylerenderer_amd import model, synth, train
    from test_model_cpu import noise_list

    size = tag_size
    sdim, nmlp, batch, zkey, nkey, salt = GWM_CASES[size]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    g = model.GeneratorWithMap(size, sdim, nmlp)
    assert sum(p.numel() for p in g.parameters()) == int(gold["n_params"])
    synth.fill_state_dict(g.state_dict(), salt=salt)
    g = g.to(dev)
    if "v" in gold.files:
        v_np, n_np, tri_np = gold["v"], gold["nrm"], gold["tri"].astype(np.int64)
    else:
        v0, tri_np = synth.face_sized_mesh()
        v_np = synth.random_poses(v0, batch, seed=9)
        n_np = synth.vertex_normals(v_np, tri_np)
        digest = np.array([float(np.abs(v_np).sum()), float(np.abs(n_np).sum()), float(tri_np.sum())], np.float64)
        assert np.array_equal(digest, gold["mesh_digest"]), (digest, gold["mesh_digest"])
    v, n = T(v_np).requires_grad_(), T(n_np).requires_grad_()
    tri = T(tri_np)
    z = T(synth.det_normal((batch, sdim), zkey))
    noise = [x.to(dev) for x in noise_list(g, nkey)]
    forcer = KinkForcer(g, gold) if "kink_keys" in gold.files else None
    img, lat, maps = g([z], (v, n, tri), return_normals=True, return_latents=True, noise=noise)
    meas = {}
    if forcer is not None:
        forcer.close()
        assert len(forcer.stats) == len(gold["kink_keys"])
        assert not forcer.unexplained(), forcer.unexplained()
        meas["forced"] = forcer.disagreements()
    step = img.shape[-1] // gold["image"].shape[-1]
    meas["img"] = rel_err(img.detach().cpu().numpy()[:, :, ::step, ::step], gold["image"])
    assert meas["img"] < tol_img, meas
    for i, m in enumerate(maps):
        k = int(gold["normmap_step_%d" % i]) if ("normmap_step_%d" % i) in gold.files else 1
        assert np.abs(m.detach().cpu().numpy()[..., ::k, ::k] - gold["normmap_%d" % i]).max() <= 2e-7
    proj = T(synth.det_normal(tuple(img.shape), zkey + 4))
    params = dict(g.named_parameters())
    grads = torch.autograd.grad((img * proj).sum(), list(params.values()) + [v, n], allow_unused=True,
                                retain_graph=True)
    got = {k: x for k, x in zip(params, grads[:-2]) if x is not None}
    assert sorted(k for k, x in zip(params, grads[:-2]) if x is None) == list(gold["unused"])
    f64 = (lambda k: gold[k] if k in gold.files else None)  # noqa: E731  (256^2: float64 truth, see check_grad_samples)
    meas["g1"] = check_grad_samples(got, gold["grad_names"], gold["grad_samples"], gold["grad_sample_offsets"], tol_g1,
                                    truth=f64("grad_samples_f64"))
    meas["gv"] = rel_err(grads[-2].cpu().numpy(), gold["grad_v"])
    meas["gn"] = rel_err(grads[-1].cpu().numpy(), gold["grad_nrm"])
    assert meas["gv"] < tol_mesh1 and meas["gn"] < tol_mesh1, meas
    # the regulariser the training step evaluates (reference train.py:340-347)
    if "pl_probe" in gold.files:
        probe = T(gold["pl_probe"])
    else:
        torch.manual_seed(int(gold["pl_probe_seed"]))
        probe = torch.randn(tuple(img.shape))
        got_digest = np.array([float(probe.double().sum()), float(probe.double().abs().sum())])
        assert np.array_equal(got_digest, gold["pl_probe_digest"]), "torch's CPU generator drew a different probe"
        probe = probe.to(dev)
    pen, mean, lengths = train.g_path_regularize(img, [lat] + list(maps), torch.tensor(0.25, device=dev), noise=probe)
    assert rel_err(lengths.detach().cpu().numpy(), gold["pl_lengths"]) < 1e-4
    assert abs(float(pen) - float(gold["pl_penalty"])) < 2e-4 * abs(float(gold["pl_penalty"]))
    assert abs(float(mean) - float(gold["pl_mean"])) < 1e-5 * abs(float(gold["pl_mean"]))
    g.zero_grad()
    v.grad = n.grad = None
    (2.0 * 4 * pen + 0 * img[0, 0, 0, 0]).backward()
    got = {k: p.grad for k, p in g.named_parameters() if p.grad is not None}
    meas["g2"] = check_grad_samples(got, gold["pl_grad_names"], gold["pl_grad_samples"],
                                    gold["pl_grad_sample_offsets"], tol_g2, truth=f64("pl_grad_samples_f64"))
    meas["gv2"] = rel_err(v.grad.cpu().numpy(), gold["pl_grad_v"])
    meas["gn2"] = rel_err(n.grad.cpu().numpy(), gold["pl_grad_nrm"])
    assert meas["gv2"] < tol_mesh2 and meas["gn2"] < tol_mesh2, meas
    return meas


class KinkForcer:
    """Forward hooks that put the LeakyReLU outputs recorded by oracle/make_golden.KinkRecorder on the REFERENCE's side
    of the kink and prove that the two sign patterns are then identical.

    For every recorded layer (the module that returns the activation output: StyledConv / StyledMapConv / ConvLayer /
    EqualLinear) the fixture holds the flat indices of the outputs whose pre-activation is within tau * max|x| of zero
    in the reference, the reference's side of each, and the per-channel count of positive outputs.  The hook
      * compares the product's side of the flagged elements with the reference's and — `force` — rewrites the ones
        that disagree to +-1e-30 (through `.data`: same storage, the value the backward kernels take the mask from;
        the forward value moves by < tau of the layer's scale at a few dozen elements of 1e7),
      * then compares the per-channel positive counts: equal counts after forcing = no sign disagreement OUTSIDE the
        flagged band (a flip there would need an activation error above tau, three orders above the measured one).
    stats[layer] = (flagged, disagreeing among them, channels whose count still differs)."""

    def __init__(self, net, gold, prefix="kink", force=True):
        import torch

        self.stats = {}
        self.hooks = []
        self.force = force
        mods = dict(net.named_modules())
        for i, key in enumerate(str(k) for k in gold[prefix + "_keys"]):
            idx = torch.from_numpy(gold["%s_idx_%d" % (prefix, i)].astype(np.int64))
            pos = torch.from_numpy(np.unpackbits(gold["%s_pos_%d" % (prefix, i)])[:idx.numel()].astype(bool))
            cnt = torch.from_numpy(gold["%s_cnt_%d" % (prefix, i)].astype(np.int64))
            shape = tuple(int(v) for v in gold["%s_shape_%d" % (prefix, i)])
            self.hooks.append(mods[key].register_forward_hook(
                lambda mod, inp, out, a=(key, idx, pos, cnt, shape): self._hook(out, *a)))

    def _hook(self, out, key, idx, pos, cnt, shape):
        import torch

        y = out.data
        assert tuple(y.shape) == shape and y.is_contiguous(), (key, tuple(y.shape), shape)
        flat = y.view(-1)
        idx, pos, cnt = idx.to(y.device), pos.to(y.device), cnt.to(y.device)
        dis = (flat[idx] > 0) != pos
        n_dis = int(dis.sum())
        if n_dis and self.force:
            flat[idx[dis]] = torch.where(pos[dis], 1e-30, -1e-30).to(flat.dtype)
        dims = [d for d in range(y.dim()) if d != 1]
        bad = int(((y > 0).sum(dims) != cnt).sum())
        prev = self.stats.get(key, (0, 0, 0))
        self.stats[key] = (int(idx.numel()), prev[1] + n_dis, prev[2] + bad)

    def close(self):
        for h in self.hooks:
            h.remove()

    def disagreements(self):
        return sum(v[1] for v in self.stats.values())

    def unexplained(self):
        return {k: v for k, v in self.stats.items() if v[2]}


def run_discriminator_case(gold, size, dev, tol_y, tol_g1, tol_g2):
    """Discriminator(size), batch 4, against tests/golden/discriminator_s<size>.npz (oracle/make_golden._disc_case):
    logits, first-order gradients (parameters sampled, input sub-sampled), one weighted R1 evaluation (reference
    train.py:110-114, 281-289) with its double-backward gradients; LeakyReLU kinks controlled by KinkForcer.  On CPU
    tensors the mask comes from the saved pre-activation, which the hook cannot reach: there a disagreement is only
    counted (the CPU path is bit-identical to the reference on the authoring host, where the CPU suite runs)."""
    import torch

    from stylerenderer_amd import model, synth, train

    d = model.Discriminator(size)
    assert sum(p.numel() for p in d.parameters()) == int(gold["n_params"])
    synth.fill_state_dict(d.state_dict(), salt=61)
    d = d.to(dev)
    sub = int(gold["sub"])
    x = torch.from_numpy(synth.det_normal((4, 3, size, size), 62 + size)).to(dev).requires_grad_()
    forcer = KinkForcer(d, gold)
    y = d(x)
    meas = {"y": rel_err(y.detach().cpu().numpy(), gold["y"])}
    assert meas["y"] < tol_y, meas
    assert not forcer.unexplained(), forcer.unexplained()
    params = dict(d.named_parameters())
    grads = torch.autograd.grad(y.sum(), list(params.values()) + [x], retain_graph=True)
    f64 = (lambda k: gold[k] if k in gold.files else None)  # noqa: E731  (256^2: float64 truth, see check_grad_samples)
    meas["g1"] = check_grad_samples(dict(zip(params, grads[:-1])), gold["grad_names"], gold["grad_samples"],
                                    gold["grad_sample_offsets"], tol_g1, truth=f64("grad_samples_f64"))
    meas["gx"] = rel_err(grads[-1].cpu().numpy()[:, :, ::sub, ::sub], gold["gx"])
    assert meas["gx"] < tol_g1, meas
    # R1 the way the step evaluates it: a second forward, d_r1_loss, weighted backward (a double backward)
    xr = x.detach().clone().requires_grad_(True)
    pred = d(xr)
    forcer.close()
    assert not forcer.unexplained(), forcer.unexplained()
    r1 = train.d_r1_loss(pred, xr)
    meas["r1"] = abs(float(r1) - float(gold["r1"])) / abs(float(gold["r1"]))
    assert meas["r1"] < tol_g1, meas
    d.zero_grad()
    (10.0 / 2 * r1 * 16 + 0 * pred[0]).backward()
    got = {n: p.grad for n, p in d.named_parameters() if p.grad is not None}
    meas["g2"] = check_grad_samples(got, gold["r1_grad_names"], gold["r1_grad_samples"], gold["r1_grad_sample_offsets"],
                                    tol_g2, truth=f64("r1_grad_samples_f64"))
    meas["forced"] = forcer.disagreements()
    return meas
