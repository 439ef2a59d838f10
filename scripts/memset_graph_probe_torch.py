#!/usr/bin/env python3
"""GPU box: the memset-node question of scripts/memset_graph_probe.cpp inside the torch process (torch's bundled HIP
runtime), and the pieces of torch's multi-block reduction one by one."""
import ctypes

import torch

dev = torch.device("cuda")
hip = ctypes.CDLL("libamdhip64.so")
ver = ctypes.c_int()
hip.hipRuntimeGetVersion(ctypes.byref(ver))
print("HIP runtime in this process:", ver.value, "torch", torch.__version__, torch.version.hip)
for nbytes in (4, 16, 256, 4096):
    n = max(1, nbytes // 4)
    sem = torch.zeros(n, dtype=torch.int32, device=dev)
    out = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        st = torch.cuda.current_stream().cuda_stream
        sem.add_(1)
        rc = hip.hipMemsetAsync(ctypes.c_void_p(sem.data_ptr()), 0, ctypes.c_size_t(nbytes), ctypes.c_void_p(st))
        sem.add_(1)
        out.copy_(sem)
    res = []
    for r in range(4):
        g.replay()
        torch.cuda.synchronize()
        res.append((int(out.min()), int(out.max())))
    print("memset node %5d bytes (rc %d): %s" % (nbytes, rc, res))
