#!/usr/bin/env python3
"""GPU box: do torch's multi-block reductions (semaphore + staging buffer, memset node) replay correctly in a hipGraph?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylerenderer_amd import graphs  # noqa: E402

dev = torch.device("cuda")
cases = [("sum(0,2,3) [4,3,256,256]", (4, 3, 256, 256), (0, 2, 3)), ("sum(1) [2,512,32,32]", (2, 512, 32, 32), (1,)),
         ("sum(1) [2,512,16,16]", (2, 512, 16, 16), (1,)), ("sum(1) [2,512,64,64]", (2, 512, 64, 64), (1,)),
         ("sum(1) [2,512,8,8]", (2, 512, 8, 8), (1,)), ("sum(2,3) [2,512,64,64]", (2, 512, 64, 64), (2, 3)),
         ("sum() [2,512,64,64]", (2, 512, 64, 64), None), ("sum(0) [2,14,512]", (2, 14, 512), (0,))]
for tag, shape, dims in cases:
    x = torch.randn(shape, device=dev)
    fn = (lambda t: t.sum()) if dims is None else (lambda t, d=dims: t.sum(d))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            y = fn(x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for mode in ("plain torch capture", "graphs.capture (memset nodes -> fill kernels)"):
        hold = {}

        def body():
            junk = torch.empty(1 << 20, device=dev).normal_()          # other work around it, as in a real phase
            y = fn(x * 1.0)
            hold["z"] = y * 2.0 + junk[:1].sum() * 0
            hold["junk2"] = torch.empty(1 << 20, device=dev).normal_()

        if mode.startswith("plain"):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body()
            extra = ""
        else:
            g = graphs.capture(body)
            extra = " (%d memset nodes replaced)" % g.memset_nodes_replaced
        res = []
        for rep in range(4):
            x.normal_()
            want = fn(x) * 2.0
            g.replay()
            torch.cuda.synchronize()
            res.append(float((hold["z"] - want).abs().max() / want.abs().max()))
        print("%-26s %-48s rel err per replay: %s%s" % (tag, mode, ["%.1e" % r for r in res], extra))
