#!/usr/bin/env python3
"""GPU box: find reads of uninitialised memory.  Every torch.empty / empty_like / new_empty issued from Python (the
operator wrappers' outputs and scratch buffers) is filled with NaN (or a huge value) before use; a kernel that does not
overwrite all of its output, or a scratch buffer that is read before it is written, then shows up as NaN / garbage in
the phase's gradients.  usage: python scripts/poison_probe.py path,g [size] [fill=nan|big]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import graph_train, train  # noqa: E402

phases = sys.argv[1].split(",") if len(sys.argv) > 1 else ["path"]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
fill = float("nan") if (len(sys.argv) <= 3 or sys.argv[3] == "nan") else 1e30
dev = torch.device("cuda")
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=size, latent=512, n_mlp=8, channel_multiplier=2, use_mesh=True, device=dev, seed=0,
                                batch=4, mesh_vertices=faces.model.dim[2] // 3, capture=False)
data = train.SyntheticImages(16, size, dev)
tr.step(data.batch(4), faces=faces)

_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
sites = {}


def note(t):
    if t.is_cuda and t.is_floating_point() and t.numel():
        t.fill_(fill)
    elif t.is_cuda and t.numel() and t.dtype in (torch.int32, torch.int64, torch.uint8):
        t.fill_(0x5A if t.dtype == torch.uint8 else 0x5A5A5A5A)
    return t


torch.empty = lambda *a, **k: note(_empty(*a, **k))
torch.empty_like = lambda *a, **k: note(_empty_like(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: note(_new_empty(self, *a, **k))

names_g = [n for n, _ in tr.generator.named_parameters() if n not in tr.frozen]
names_d = [n for n, _ in tr.discriminator.named_parameters()]
for phase in phases:
    flat = tr.flat_g if phase in ("g", "path") else tr.flat_d
    params = tr.g_params if phase in ("g", "path") else tr.d_params
    opt = tr.g_optim if phase in ("g", "path") else tr.d_optim
    names = names_g if phase in ("g", "path") else names_d
    mpl = tr.mean_path_length.clone()
    state = torch.cuda.get_rng_state(dev)
    torch.empty, torch.empty_like, torch.Tensor.new_empty = _empty, _empty_like, _new_empty
    flat.zero_()
    tr._bodies()[phase]()
    clean = flat.clone()
    tr.mean_path_length.copy_(mpl)
    torch.cuda.set_rng_state(state, dev)
    torch.empty = lambda *a, **k: note(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: note(_empty_like(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: note(_new_empty(self, *a, **k))
    flat.zero_()
    tr._bodies()[phase]()
    torch.cuda.synchronize()
    bad = []
    for n, p, o in zip(names, params, opt.offs):
        a, b = clean[o:o + p.numel()], flat[o:o + p.numel()]
        if not torch.equal(a, b):
            bad.append((n, int((~torch.isfinite(b)).sum()), float((a - b).nan_to_num(0).abs().max()), float(a.abs().max())))
    print("=== phase %s: %d of %d parameter gradients change under poisoned empties; losses %s" % (
        phase, len(bad), len(names), {k: float(v) for k, v in tr.s_loss.items() if k in ("g", "d", "path", "r1")}))
    for row in bad[:60]:
        print("   %-44s nonfinite %8d  maxdiff %.3e scale %.3e" % row)
    tr.mean_path_length.copy_(mpl)
