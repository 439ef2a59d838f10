"""GPU box: the rasterizer leg of bench.py alone (BASELINE config[3]) — forward / forward+backward / backward ms.
usage: scripts/raster_bench.py [iters] [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import bench  # noqa: E402

out = bench.raster_leg(torch.device("cuda", 0), 1, batch=int(sys.argv[2]) if len(sys.argv) > 2 else 64,
                       iters=int(sys.argv[1]) if len(sys.argv) > 1 else 20, cpu_baseline=False)
print(json.dumps({k: out[k] for k in ("fwd_ms", "fwd_api_ms", "fwd_bwd_ms", "bwd_ms", "fwd_mtri_s", "fwd_bwd_mtri_s")}))
