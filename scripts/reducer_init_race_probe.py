#!/usr/bin/env python3
"""GPU box: does a wait kernel queued right after BucketedGradReducer() see the zeroed signal words?  The words are filled
on the current stream and read on the communication stream; `noedge` removes the stream edge the constructor adds.
usage: python scripts/reducer_init_race_probe.py [noedge] [rounds=300]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stylerenderer_amd import _lib  # noqa: E402
from stylerenderer_amd import distributed as sr_dist  # noqa: E402

noedge = "noedge" in sys.argv
rounds = next((int(a) for a in sys.argv[1:] if a.isdigit()), 300)
os.environ["SR_SIGNAL_TIMEOUT_S"] = "0.002"
dev = torch.device("cuda", 0)
ps = [torch.nn.Parameter(torch.zeros(64, device=dev)) for _ in range(4)]
flat = torch.zeros(256, device=dev)
views = [flat[i * 64:(i + 1) * 64] for i in range(4)]
L = _lib.lib()
real_wait = torch.cuda.Stream.wait_stream
missed = 0
big = torch.empty(64 << 20, device=dev)
for r in range(rounds):
    junk = torch.full((2,), 7, dtype=torch.int32, device=dev)      # the block the reducer's words will reuse
    torch.cuda.synchronize()
    del junk
    big.normal_()                                                  # keeps the current stream busy for a while
    if noedge:
        torch.cuda.Stream.wait_stream = lambda self, other: None
    red = sr_dist.BucketedGradReducer(ps, views, [0, 64, 128, 192], flat, world=1, n_buckets=2, force=True)
    torch.cuda.Stream.wait_stream = real_wait
    _lib.check(L.sr_signal_wait_timeout(red.counters.data_ptr() + 4, 1, red.timeout_us, red.status.data_ptr() + 4, 2,
                                        red.comm.cuda_stream), "wait")
    torch.cuda.synchronize()
    if not bool(red.status.any()):
        missed += 1                                                # the wait returned without timing out: stale word
    red.status.zero_()
print("%s: %d of %d waits saw a stale (non-zero) word" % ("no edge" if noedge else "with edge", missed, rounds))
