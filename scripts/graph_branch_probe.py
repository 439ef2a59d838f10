"""GPU box: do parallel branches of a captured hipGraph (fork / join through a side stream inside the capture) run
concurrently at replay on this runtime?  A = chain of `n_small` tiny kernels, B = chain of `n_big` GPU-filling kernels.
Prints replay time of A;B in one chain vs A || B."""
import sys
import time

import torch

dev = torch.device("cuda")
n_small, n_big = int(sys.argv[1]), int(sys.argv[2])
small = [torch.zeros(256, device=dev) for _ in range(4)]
big_a = torch.randn(8192, 8192, device=dev)
big_b = torch.randn(8192, 8192, device=dev)
big_c = torch.empty(8192, 8192, device=dev)


def chain_small():
    for i in range(n_small):
        small[i % 4].add_(1.0)


def chain_big():
    for _ in range(n_big):
        torch.mm(big_a, big_b, out=big_c)


def serial():
    chain_small()
    chain_big()


side = torch.cuda.Stream()


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain_small()
    chain_big()
    main.wait_stream(side)


def timed(body):
    for _ in range(2):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3


print("small chain alone  %.3f ms" % timed(chain_small))
print("big chain alone    %.3f ms" % timed(chain_big))
print("serial             %.3f ms" % timed(serial))
print("forked (A || B)    %.3f ms" % timed(forked))
