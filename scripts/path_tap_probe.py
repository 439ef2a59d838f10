#!/usr/bin/env python3
"""GPU box: where does the hipGraph replay of the path-length phase diverge from the eager body?  Forward hooks copy
every module output of the generator into persistent tap buffers (the copies are captured too); after a replay the
taps are compared, in forward order, with those of the eager run from the same state."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
phase = sys.argv[2] if len(sys.argv) > 2 else "path"
dev = torch.device("cuda")
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=size, latent=512, n_mlp=8, channel_multiplier=2, use_mesh=True, device=dev, seed=0,
                                batch=4, mesh_vertices=faces.model.dim[2] // 3, capture=False)
data = train.SyntheticImages(16, size, dev)
for _ in range(2):
    tr.step(data.batch(4), faces=faces)

taps, order = {}, []


def tap(name, t):
    if not torch.is_tensor(t) or not t.is_floating_point():
        return
    if name not in taps:
        taps[name] = torch.zeros_like(t)
        order.append(name)
    taps[name].copy_(t.detach())


def fwd_hook(name):
    def hook(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for k, o in enumerate(outs):
            if isinstance(o, (tuple, list)):
                for j, oo in enumerate(o):
                    tap("%s.out%d.%d" % (name, k, j), oo)
            else:
                tap("%s.out%d" % (name, k), o)
            if torch.is_tensor(o) and o.requires_grad and os.environ.get("TAP_GRADS", "1") == "1":
                cnt = {"n": 0}

                def ghook(g, name=name, k=k, cnt=cnt):
                    tap("%s.out%d.grad#%d" % (name, k, cnt["n"]), g)
                    cnt["n"] += 1

                o.register_hook(ghook)
    return hook


for name, mod in tr.generator.named_modules():
    if name and name.count(".") <= 1:
        mod.register_forward_hook(fwd_hook(name))

body = tr._bodies()[phase]
mpl = tr.mean_path_length.clone()
state = torch.cuda.get_rng_state(dev)


def reset():
    tr.mean_path_length.copy_(mpl)
    torch.cuda.set_rng_state(state, dev)
    tr.flat_g.zero_()
    for t in taps.values():
        t.zero_()


reset()
body()
torch.cuda.synchronize()
ref = {k: v.clone() for k, v in taps.items()}
ref_flat = tr.flat_g.clone()
print("taps:", len(order))
reset()
body()
torch.cuda.synchronize()
print("eager again: %d taps differ" % sum(not torch.equal(taps[k], ref[k]) for k in order))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    reset()
    body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
reset()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    body()
for rep in range(3):
    reset()
    graph.replay()
    torch.cuda.synchronize()
    bad = [(k, float((taps[k] - ref[k]).abs().max()), float(ref[k].abs().max())) for k in order
           if not torch.equal(taps[k], ref[k])]
    print("replay %d: %d of %d taps differ; flat max diff %.3e" % (rep, len(bad), len(order),
                                                                  float((tr.flat_g - ref_flat).abs().max())))
    for row in bad[:25]:
        print("    %-50s diff %.3e scale %.3e" % row)
