#!/usr/bin/env python3
"""GPU box: which flat-gradient slots differ between an eager phase body and its hipGraph replay (same RNG state)?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from test_graph_trainer_gpu import full_size_trainer  # noqa: E402

phases = sys.argv[1].split(",") if len(sys.argv) > 1 else ["g"]
tr, faces, data = full_size_trainer()
for _ in range(3):
    tr.step(data.batch(4), faces=faces)
dev = tr.device
for phase in phases:
  print("=== phase", phase)
  flat = tr.flat_g if phase in ("g", "path") else tr.flat_d
  params = tr.g_params if phase in ("g", "path") else tr.d_params
  opt = tr.g_optim if phase in ("g", "path") else tr.d_optim
  names = ([n for n, _ in tr.generator.named_parameters() if n not in tr.frozen] if phase in ("g", "path")
           else [n for n, _ in tr.discriminator.named_parameters()])
  runs = {}
  mpl = tr.mean_path_length.clone()
  state = torch.cuda.get_rng_state(dev)
  for key in ("eager1", "graph1", "eager2", "graph2"):
      tr.mean_path_length.copy_(mpl)
      torch.cuda.set_rng_state(state, dev)
      flat.zero_()
      if key.startswith("eager"):
          tr._bodies()[phase]()
      else:
          tr.graphs[phase].replay()
      torch.cuda.synchronize()
      runs[key] = (flat.clone(), {k: float(v) for k, v in tr.s_loss.items()})
  for a, b in (("eager1", "eager2"), ("graph1", "graph2"), ("eager1", "graph1")):
      fa, fb = runs[a][0], runs[b][0]
      print("%s vs %s: max|diff| %.3e of %.3e; losses %s | %s" % (a, b, float((fa - fb).abs().max()), float(fa.abs().max()),
            {k: round(v, 5) for k, v in runs[a][1].items() if k in ("g", "d", "path", "r1")},
            {k: round(v, 5) for k, v in runs[b][1].items() if k in ("g", "d", "path", "r1")}))
      if a == "eager1" and b == "graph1":
          bad = []
          for n, p, o in zip(names, params, opt.offs):
              d = float((fa[o:o + p.numel()] - fb[o:o + p.numel()]).abs().max())
              s = float(fa[o:o + p.numel()].abs().max())
              if d > 1e-5 * max(s, 1e-12):
                  bad.append((n, d, s))
          print("%d of %d parameters differ" % (len(bad), len(names)))
          for row in bad[:40]:
              print("   %-44s diff %.3e scale %.3e" % row)
