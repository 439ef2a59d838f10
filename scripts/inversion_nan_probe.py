#!/usr/bin/env python3
"""GPU box: where does the full-size inversion loop go non-finite?  Eager iterations with per-step diagnostics."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from test_inversion_gpu import full_size_setup  # noqa: E402

inv = full_size_setup(False)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    inv.w.grad = None
    inv.pose.grad = None
    img = inv.render()
    feats = inv.perceptual.features(img)
    zero_px = [int((f.abs().sum(1) == 0).sum()) for f in feats]
    d = inv.perceptual.distance_to(inv.target_feats, img).mean()
    mse = torch.mean((img - inv.target) ** 2)
    value = d + inv.pixel_weight * mse
    value.backward()
    print("step %2d  lpips %.5f  mse %.5f  |img| %.3f  zero-feature pixels %s  gw finite %s gpose finite %s |gw| %.3e" % (
        i, float(d), float(mse), float(img.abs().max()), zero_px, bool(torch.isfinite(inv.w.grad).all()),
        bool(torch.isfinite(inv.pose.grad).all()), float(inv.w.grad.abs().max())), flush=True)
    if not torch.isfinite(inv.w.grad).all():
        break
    inv.optim.step()
