import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from test_bankmm_gpu import _generator, _linear_activations, DEV
from stylerenderer_amd import train
g = _linear_activations(_generator())
names = ["w"] + [n for n, p in g.named_parameters() if p.requires_grad]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
z = torch.randn(batch, 512, device=DEV, generator=torch.Generator(DEV).manual_seed(5))
noise = [n.detach() for n in g.make_noise()]
params = [p for p in g.parameters() if p.requires_grad]
def run(mode, which="both"):
    os.environ["SR_STYLE_BANK"] = mode
    os.environ["SR_STRICT_NATIVE"] = "0"
    w = g.style(z).unsqueeze(1).repeat(1, g.n_latent, 1).detach().requires_grad_(True)
    img, _ = g([w], input_is_latent=True, noise=noise)
    probe = torch.randn(img.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(11))
    loss, mean, lengths = train.g_path_regularize(img, w, torch.zeros((), device=DEV), noise=probe)
    obj = {"path": loss, "img": img.square().mean(), "both": loss + img.square().mean()}[which]
    return [img.detach()] + list(torch.autograd.grad(obj, [w] + params, allow_unused=True))
for mode in sys.argv[2:] or ["1", "0"]:
    for which in ("img", "path"):
        runs = [run(mode, which) for _ in range(4)]
        for k in range(1, 4):
            bad = [(names[i - 1] if i else "image", float((a - b).abs().max())) for i, (a, b) in enumerate(zip(runs[0], runs[k]))
                   if a is not None and not torch.equal(a, b)]
            print("mode %s %s run 0 vs %d: %d tensors differ %s" % (mode, which, k, len(bad), bad[:6]))
print("--- the order of the test: 1, stacked, 0, 1")
seq = [run(m) for m in ("1", "stacked", "0", "1", "1")]
for k in (3, 4):
    bad = [(names[i - 1] if i else "image", float((a - b).abs().max()), float(b.abs().max())) for i, (a, b) in enumerate(zip(seq[0], seq[k]))
           if a is not None and not torch.equal(a, b)]
    print("first vs run %d: %d tensors differ %s" % (k, len(bad), bad[:8]))
bad = [(names[i - 1] if i else "image", float((a - b).abs().max())) for i, (a, b) in enumerate(zip(seq[3], seq[4])) if a is not None and not torch.equal(a, b)]
print("run 3 vs 4: %d differ %s" % (len(bad), bad[:5]))
