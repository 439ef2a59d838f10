"""hipGraph capture for the training / inversion loops (torch.cuda.CUDAGraph on ROCm) with the repair this stack needs.

PyTorch 2.10+rocm7.0 ships the HIP 7.0.51831 runtime, whose captured MEMSET nodes write a corrupted value from the
second replay on (scripts/memset_graph_probe_torch.py).  torch's multi-block reductions (`sum`, `mean`, autograd's
`sum_to_size`, ... whenever one output needs more than one workgroup) zero their semaphores with `hipMemsetAsync`, so
inside a replayed graph they returned garbage: scripts/graph_reduce_probe.py, and in this repository the path-length
phase of the training iteration and the pixel loss of the inversion loop (found in round 3 by the full-size
graph-vs-eager tests; round 2 compared only the discriminator phase, which has no such reduction).

`capture(body)` therefore keeps the captured hipGraph_t (`keep_graph=True`), lets the C ABI replace every memset node
by a fill-kernel node with the same edges (`sr_graph_replace_memset_nodes`, csrc/capi.hip) and only then instantiates.
Replay stays torch's (`graph.replay()`), so the Philox offset bookkeeping of captured random ops is untouched.
"""
import ctypes
import weakref

import torch

from . import _lib
from .op import _dispatch


def capture(body, pool=None):
    """Captures `body()` on torch's capture stream; returns the repaired, instantiated graph.  `graph.memset_nodes_replaced`
    holds the number of memset nodes that were rewritten."""
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    kw = {"pool": pool} if pool is not None else {}
    # thread_local: only THIS thread's calls are policed during capture.  Under the default (global) mode the RCCL
    # watchdog thread's routine hipEventQuery on an earlier collective aborts the process with "operation not
    # permitted when stream is capturing".
    # derived-tensor cache entries (flipped FIR taps, incidence lists, frozen weight adjoints) touched by the capture
    # stay pinned exactly as long as this graph object lives
    scope = _dispatch.PinScope()
    with _dispatch.pin_scope(scope), torch.cuda.graph(graph, capture_error_mode="thread_local", **kw):
        body()
    weakref.finalize(graph, scope.release)
    graph.pin_scope = scope
    n = ctypes.c_int(0)
    raw = graph.raw_cuda_graph()
    _lib.check(_lib.lib().sr_graph_replace_memset_nodes(ctypes.c_void_p(int(raw)), ctypes.byref(n)),
               "sr_graph_replace_memset_nodes")
    kernels, total = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.lib().sr_graph_node_count(ctypes.c_void_p(int(raw)), ctypes.byref(kernels), ctypes.byref(total)),
               "sr_graph_node_count")
    graph.instantiate()
    graph.memset_nodes_replaced = int(n.value)
    graph.kernel_nodes, graph.nodes = int(kernels.value), int(total.value)       # launches per replay
    return graph
