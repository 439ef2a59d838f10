#!/usr/bin/env python3
"""GPU box, torch process (bundled HIP runtime): sr_signal_bump as a node inside a captured graph + sr_signal_wait on
another stream; does the waiting stream start in the MIDDLE of the replay?"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylerenderer_amd import _lib, graphs  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda")
counter = torch.zeros(1, dtype=torch.int32, device=dev)
comm = torch.cuda.Stream()
x = torch.zeros(1 << 20, device=dev)
a = torch.randn(8192, 8192, device=dev)
rcs = []


def busy():                       # ~2 ms of matrix products
    for _ in range(3):
        torch.mm(a, a)


def body():
    busy()
    x.add_(1)
    rcs.append(L.sr_signal_bump(counter.data_ptr(), _lib.current_stream(dev)))
    busy()
    x.add_(1)


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    busy()
torch.cuda.synchronize()
g = graphs.capture(body)
print("bump rc in capture:", rcs, "counter after capture:", int(counter))
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    g.replay()
    e1.record()
    rc = L.sr_signal_wait(counter.data_ptr(), rep + 1, comm.cuda_stream)
    with torch.cuda.stream(comm):
        y = x[:16] * 2
        ec.record()
    torch.cuda.synchronize()
    print("replay %d: graph %.2f ms; consumer finished %.2f ms after graph start (wait rc %d) -> %s" % (
        rep, e0.elapsed_time(e1), e0.elapsed_time(ec), rc,
        "OVERLAP" if e0.elapsed_time(ec) < 0.75 * e0.elapsed_time(e1) else "no overlap"))
