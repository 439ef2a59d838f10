#!/usr/bin/env python3
"""GPU box: BASELINE config[4] loop alone (for rocprofv3 traces).  usage: python scripts/inversion_probe.py [steps=60]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t0 = time.perf_counter()
out = bench.inversion_leg(torch.device("cuda", 0), steps, 256)
print({k: out[k] for k in ("value", "eager_steps_per_s", "loss_first", "loss_last")}, "%.1f s" % (time.perf_counter() - t0))
