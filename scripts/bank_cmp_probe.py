import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from test_bankmm_gpu import _generator, _linear_activations, DEV
from stylerenderer_amd import train
g = _linear_activations(_generator()) if os.environ.get('LINEAR', '1') == '1' else _generator()
for batch in (1, 4):
    z = torch.randn(batch, 512, device=DEV, generator=torch.Generator(DEV).manual_seed(5))
    noise = [n.detach() for n in g.make_noise()]
    params = [p for p in g.parameters() if p.requires_grad]
    def run(mode, which):
        os.environ["SR_STYLE_BANK"] = mode
        os.environ["SR_STRICT_NATIVE"] = "0"
        w = g.style(z).unsqueeze(1).repeat(1, g.n_latent, 1).detach().requires_grad_(True)
        img, _ = g([w], input_is_latent=True, noise=noise)
        probe = torch.randn(img.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(11))
        loss, mean, lengths = train.g_path_regularize(img, w, torch.zeros((), device=DEV), noise=probe)
        obj = loss if which == "path" else img.square().mean()
        return torch.autograd.grad(obj, [w] + params, allow_unused=True)
    for which in ("path", "img"):
        res = {m: run(m, which) for m in ("1", "stacked", "0")}
        for a, b in (("1", "stacked"), ("1", "0"), ("stacked", "0")):
            worst = 0.0; wi = -1
            for i, (x, y) in enumerate(zip(res[a], res[b])):
                if x is None: continue
                d = float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30)
                if d > worst: worst, wi = d, i
            print("batch %d %-4s %-8s vs %-8s worst rel-to-max diff %.2e (tensor %d)  latent: %.2e" % (
                batch, which, a, b, worst, wi, float((res[a][0]-res[b][0]).abs().max())/float(res[b][0].abs().max())))
