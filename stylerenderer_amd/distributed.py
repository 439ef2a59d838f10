"""Data-parallel plumbing: one process per GPU, `torch.distributed` over RCCL (backend "nccl" on
PyTorch-ROCm) across the node's xGMI mesh; "gloo" for the CPU tests.

Same helper names as the reference's distributed.py (get_rank, get_world_size, synchronize,
reduce_sum, reduce_loss_dict, initialize, construct_ddp), different behaviour where the
reference serialises the step:
  * the reference all-reduces 7 logging scalars one by one, each followed by `.item()`
    (reference train.py:184-189, 359-360) — `reduce_scalars` packs them into ONE tensor, one
    all-reduce, one host read;
  * `construct_ddp` freezes the duplicated ToRGB tail that never receives a gradient
     instead of paying for `find_unused_parameters`, keeps
    `broadcast_buffers=False` like the reference (distributed.py:104: noise / FIR buffers are
    constant), and uses gradient-as-bucket-view with 32 MB buckets so the all-reduce of the
    125 MB of generator gradients overlaps with the remaining backward (ring bound per bucket
    2*(N-1)/N * S / 153 GB/s on the 7-link xGMI mesh).
"""
import logging
import os

import torch
from torch import distributed as dist

_log = logging.getLogger("stylerenderer_amd.distributed")


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def reduce_sum(tensor):
    """All-reduce(SUM) of a copy (reference distributed.py:21-26)."""
    if get_world_size() > 1:
        tensor = tensor.clone()
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def reduce_scalars(values, to_host=True):
    """dict name -> 0-d tensor/float  ->  dict name -> python float, averaged over ranks with ONE
    collective and ONE device->host copy.  to_host=False skips the copy (no host synchronisation:
    the step keeps enqueueing) and returns 0-d device tensors."""
    keys = sorted(values)
    if not keys:
        return {}
    ref = next((v for v in values.values() if torch.is_tensor(v)), None)
    device = ref.device if ref is not None else "cpu"
    packed = torch.stack([torch.as_tensor(values[k], dtype=torch.float32, device=device).detach().reshape(())
                          for k in keys])
    world = get_world_size()
    if world > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        packed = packed / world
    if not to_host:
        return dict(zip(keys, packed.unbind(0)))
    host = packed.cpu().tolist()
    return dict(zip(keys, host))


def reduce_loss_dict(loss_dict):
    return reduce_scalars(loss_dict)


def initialize(backend=None, seed=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run).
    Returns (rank, local_rank, world_size, device)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")
        if backend == "nccl":
            # RCCL's kernels on a high-priority stream (a hardware queue of their own, see BucketedGradReducer)
            opts = None
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:       # a torch build without the option: default streams
                opts = None
            dist.init_process_group(backend=backend, device_id=device, pg_options=opts)
        else:
            dist.init_process_group(backend=backend)
        synchronize()
    if seed is not None:
        torch.manual_seed(seed + rank)
    return rank, local_rank, world, device


def freeze_unused_tail(generator):
    """The second half of `generator.to_rgbs` is registered but never used."""
    mods = list(generator.to_rgbs)
    for m in mods[len(mods) // 2:]:
        for p in m.parameters():
            p.requires_grad_(False)


def construct_ddp(model, device=None, bucket_cap_mb=32):
    if get_world_size() <= 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP

    if hasattr(model, "to_rgbs"):
        freeze_unused_tail(model)
    on_gpu = device is not None and torch.device(device).type == "cuda"
    return DDP(model, device_ids=[torch.device(device).index] if on_gpu else None,
               broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)


def unwrap(model):
    return model.module if hasattr(model, "module") else model


class FlatGradReducer:
    """Average one flat gradient buffer over the ranks (graph_train.GraphedTrainer: one call per phase).

    Two equivalent collectives, chosen once by timing them on the real buffer:
      * all-reduce (RCCL picks ring / tree): per-link bound 2 (N-1)/N * S / 153 GB/s on a ring;
      * reduce-scatter + all-gather, in place: every GPU owns 1/N of the buffer in between, and on the fully
        connected xGMI mesh each of the 7 peer links carries 1/N of the data at once (bound 2 (S/N) / 153 GB/s).
    The decision is made from the MAX time over ranks (one tiny all-reduce), so every rank takes the same branch;
    SR_GRAD_COLLECTIVE=allreduce|rsag pins it."""

    def __init__(self, flat, world=None):
        self.world = world if world is not None else get_world_size()
        self.flat = flat
        self.mode = "none" if self.world <= 1 else "allreduce"
        self.timings = {}
        if self.world <= 1:
            return
        n = flat.numel()
        self.shardable = n % self.world == 0
        pin = os.environ.get("SR_GRAD_COLLECTIVE", "")
        if pin in ("allreduce", "rsag"):
            self.mode = pin if (pin == "allreduce" or self.shardable) else "allreduce"
        elif self.shardable and flat.is_cuda:
            try:
                self.mode = self._autotune()
            except RuntimeError as e:              # a backend without in-place reduce-scatter: every rank lands here
                self.timings = {"autotune_error": str(e)[:200]}
                self.mode = "allreduce"

    def _shard(self):
        n = self.flat.numel() // self.world
        r = get_rank()
        return self.flat[r * n:(r + 1) * n]

    def _allreduce(self):
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(self.world)

    def _rsag(self):
        shard = self._shard()
        dist.reduce_scatter_tensor(shard, self.flat, op=dist.ReduceOp.SUM)      # in place: output aliases the input
        shard.div_(self.world)
        dist.all_gather_into_tensor(self.flat, shard)

    def _autotune(self, reps=3):
        keep = self.flat.clone()
        times = {}
        for name, fn in (("allreduce", self._allreduce), ("rsag", self._rsag)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[name] = e0.elapsed_time(e1) / reps
        self.flat.copy_(keep)
        t = torch.tensor([times["allreduce"], times["rsag"]], device=self.flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                                # the same verdict on every rank
        self.timings = {"allreduce_ms": float(t[0]), "rsag_ms": float(t[1])}
        return "rsag" if float(t[1]) < 0.95 * float(t[0]) else "allreduce"

    def __call__(self):
        if self.mode == "allreduce":
            self._allreduce()
        elif self.mode == "rsag":
            self._rsag()


def arrival_order(params, run_backward):
    """Indices of `params` in the order their gradients are produced by `run_backward()` (a representative
    forward + backward); parameters it does not reach keep their relative order at the end.  Rank 0's order is
    broadcast so that every rank derives the same flat layout and buckets (torch DDP does the same when it rebuilds
    its buckets in arrival order after the first iteration)."""
    seen = []
    handles = [p.register_post_accumulate_grad_hook(lambda p, i=i: seen.append(i)) for i, p in enumerate(params)]
    try:
        run_backward()
    finally:
        for h in handles:
            h.remove()
    for p in params:
        p.grad = None
    rest = [i for i in range(len(params)) if i not in set(seen)]
    order = torch.tensor(seen + rest, dtype=torch.int64, device=params[0].device)
    if get_world_size() > 1:
        dist.broadcast(order, 0)
    return [int(i) for i in order.tolist()]


class BucketedGradReducer:
    """Averages a flat gradient buffer over the ranks WHILE the backward that fills it is still running — the
    overlap of reference distributed.py:98-105 (torch DDP bucket hooks, used for the path-length step at
    train.py:335-352) for a backward that is replayed as one hipGraph.

    The flat buffer is laid out in gradient ARRIVAL order (`arrival_order`) and cut into `n_buckets` contiguous
    slices of linearly decreasing size.  A post-accumulate hook per parameter counts arrivals; when the last gradient of a
    bucket is there, the bucket's gradients are copied into their flat views (one multi-tensor copy) and the bucket
    is SIGNALLED.  Under stream capture the signal is a one-lane kernel node in the middle of the phase graph that
    publishes the replay's EPOCH into the bucket's device word (`sr_signal_set`, include/stylerenderer_amd.h).  The
    host calls `arm()` in front of every replay (epoch += 1, written to a device scalar in stream order) and
    `issue_all()` right after it: for every bucket, `sr_signal_wait_timeout(word_k, epoch)` +
    `all_reduce(flat[lo_k:hi_k])` on the communication stream — bucket k is on the xGMI links while the graph is
    still computing the gradients of bucket k+1.  The release test is "the word has reached THIS replay's epoch": a
    replay nobody armed (a bare `graph.replay()` in a probe, an exception between replay and `issue_all()`) republishes
    an epoch already reached and cannot put the device ahead of the host's count, which is what an incrementing
    counter does (then every later wait passes at once and RCCL reduces a half-written bucket).
    A wait that is not released within SR_SIGNAL_TIMEOUT_S (default 120) stores the bucket's id into pinned host
    memory and returns; `check()` — called by arm() / issue_all() / wait() and by `wait(deadline_s=...)`'s host-side
    poll — raises with that id instead of the job hanging until an outer limit kills it.
    Eagerly (warm-up, `capture=False`, CPU / gloo) the same hooks record an ordinary event and issue the collective
    directly.
    Buckets are issued in index order on every rank, whatever order they completed in.  `wait()` makes the
    current stream (CPU: the caller) wait for all reductions — it sits in front of the optimiser step.

    No RCCL call is captured into a graph.  n_buckets = 1 degenerates to "reduce after the backward" and then uses
    FlatGradReducer (autotuned all-reduce vs reduce-scatter + all-gather); SR_GRAD_OVERLAP=0 selects that.

    Two ways to release a bucket of a REPLAYED backward (`self.release`; eager passes use events):
      "device"  the one-lane wait kernel above spins at the head of the communication stream.  It needs the
                communication stream and the replaying stream on DIFFERENT hardware queues: if HIP multiplexes them
                onto one (GPU_MAX_HW_QUEUES, other streams of the process) the waiter can sit in front of the signal it
                waits for and only its timeout ends the stall.
      "host"    the signal node stores the epoch into PINNED HOST memory (sr_signal_set_host); `issue_all()` polls
                the word from the CPU and queues the bucket's collective when it arrives — nothing spins on the device,
                no assumption about queues; the host no longer runs ahead of the replay (the optimiser graph is
                launched when the last bucket is out: ~0.1 ms of launch latency per phase).
    SR_GRAD_OVERLAP=device|host forces one; otherwise the constructor PROBES the assumption — a wait queued on the
    communication stream BEFORE the kernel that releases it is queued on the current stream; released within
    microseconds on separate queues, timed out (0.5 s) on a shared one — and picks "host" when the probe fails.
    The two modes move the same bytes through the same collectives and give bit-identical results.

    A device-side wait that times out must not let the step continue (its all-reduce used to run on a
    half-written bucket, the optimiser graph was replayed on it, peers never noticed): the wait overwrites the bucket's
    first element with NaN before it returns (sr_signal_wait_poison), the SUM carries the NaN to every rank, and the
    Adam step of an optimiser attached with `guard(optimiser)` refuses the update on every rank
    (sr_adam_flat_guarded) and raises its pinned `skipped` word, which `check()` turns into an exception on every
    rank at its next arm() / wait()."""

    def __init__(self, params, views, offs, flat, world=None, n_buckets=4, force=False, release=None):
        from . import _lib

        self.params, self.views, self.flat = list(params), list(views), flat
        self.world = world if world is not None else get_world_size()
        self.enabled = self.world > 1 or bool(force)
        self.is_cuda = flat.is_cuda
        layout = sorted(range(len(self.params)), key=lambda i: offs[i])
        total = sum(self.params[i].numel() for i in layout)
        k = max(1, min(int(n_buckets), len(layout)))
        # bucket sizes fall off linearly (k : k-1 : ... : 1, i.e. 40 / 30 / 20 / 10 % for four): what is still on the
        # wire when the backward ends is the LAST bucket, so it is the smallest
        cuts = [sum(k - j for j in range(b + 1)) / (k * (k + 1) / 2) for b in range(k)]
        self.buckets, members, acc = [], [], 0
        for pos, i in enumerate(layout):
            members.append(i)
            acc += self.params[i].numel()
            if (acc >= total * cuts[len(self.buckets)] and len(self.buckets) < k - 1) or pos == len(layout) - 1:
                lo = offs[members[0]]
                hi = offs[layout[pos + 1]] if pos + 1 < len(layout) else flat.numel()
                self.buckets.append({"lo": lo, "hi": hi, "members": members})
                members = []
        self.bucket_of = {i: b for b, bk in enumerate(self.buckets) for i in bk["members"]}
        # A producer whose LAST kernel writes a parameter's whole gradient (op.weight_prep._WPrepBwd: every convolution
        # weight, ~90 % of the bytes) may write it straight into the parameter's slot of the flat buffer: `claim` hands out
        # a fresh view of the slot once per backward (a second contribution in the same backward gets None and autograd
        # adds as usual); _flush then has nothing to copy for that parameter.  SR_GRAD_INPLACE=0 disables.
        import os

        self.claimed = set()
        if os.environ.get("SR_GRAD_INPLACE", "1") != "0":
            for i, prm in enumerate(self.params):
                prm._sr_grad_slot = (lambda i=i: self._claim(i))
        self.single = FlatGradReducer(flat, self.world) if (len(self.buckets) == 1 and self.world > 1) else None
        self.comm = None
        self.counters, self.runs, self.eager_events = None, [], []
        if self.is_cuda and self.enabled:
            self._lib = _lib
            # HIGH priority: HIP multiplexes streams of one priority over a few hardware queues (GPU_MAX_HW_QUEUES,
            # default 4).  A normal-priority communication stream that lands on the hardware queue the graph is replayed
            # on runs its wait kernel BEHIND the whole replay — correct, but nothing overlaps (seen when earlier streams
            # of the process had shifted the round-robin).  High-priority streams get queues of their own.
            self.comm = torch.cuda.Stream(device=flat.device, priority=-1)
            self.timeout_us = int(float(os.environ.get("SR_SIGNAL_TIMEOUT_S", "120")) * 1e6)
            want = release or os.environ.get("SR_GRAD_OVERLAP", "auto")
            self.release_probe = None
            if want in ("device", "host"):
                self.release = want
            else:
                self.release_probe = self._queues_independent()
                self.release = "device" if self.release_probe["independent"] else "host"
            _log.info("BucketedGradReducer rank %d: %s-released buckets (%s)", get_rank(), self.release,
                      "forced (release= / SR_GRAD_OVERLAP)" if self.release_probe is None else "queue probe %s" % self.release_probe)
            if self.release == "host":
                self.counters = torch.zeros(len(self.buckets), dtype=torch.int32).pin_memory()
                self._words = self.counters.numpy()        # the same pinned words, read without a tensor op
            else:
                self.counters = torch.zeros(len(self.buckets), dtype=torch.int32, device=flat.device)
            self.epoch = 0                               # replays announced so far (arm())
            self.epoch_dev = torch.zeros(1, dtype=torch.int32, device=flat.device)
            self.status = torch.zeros(len(self.buckets), dtype=torch.int32).pin_memory()   # written by a timed-out wait
            self.eager_events = [None] * len(self.buckets)
            # the words are zeroed on the CURRENT stream and read by wait kernels on the communication stream: without this
            # edge a wait queued right after construction can run before the fill and read whatever the cached block held
            # (a previous reducer's epochs: it returns at once — seen as a rare failure of the lost-signal test on a cold box)
            self.comm.wait_stream(torch.cuda.current_stream(flat.device))
        self.active = False
        self.guarded = []                # pinned `skipped` words of the optimisers attached with guard()
        self.pending, self.done, self.next_issue = [], [], 0
        self.works = []
        self.log = []                    # ("flush" | "issue", bucket, time.perf_counter()) of the last eager pass
        self.stamp = None                # optional callable -> float (tests); perf_counter by default
        self._handles = [p.register_post_accumulate_grad_hook(lambda p, i=i: self._arrived(i))
                         for i, p in enumerate(self.params)]

    def _queues_independent(self, timeout_s=0.5):
        """Does a kernel spinning at the head of the communication stream leave the current stream running?  The wait
        is queued FIRST, the kernel that releases it afterwards on the current stream — the order that deadlocks when
        both streams feed one hardware queue.  Both kernels are launched once before (code-object load)."""
        import time

        L, dev = self._lib.lib(), self.flat.device
        cur = torch.cuda.current_stream(dev)
        word = torch.zeros(2, dtype=torch.int32, device=dev)
        one = torch.ones(1, dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._lib.check(L.sr_signal_set(word.data_ptr() + 4, one.data_ptr(), cur.cuda_stream), "sr_signal_set")
        self.comm.wait_stream(cur)
        self._lib.check(L.sr_signal_wait_timeout(word.data_ptr() + 4, 1, 0, None, 0, self.comm.cuda_stream),
                        "sr_signal_wait_timeout")
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        self._lib.check(L.sr_signal_wait_timeout(word.data_ptr(), 1, int(timeout_s * 1e6), status.data_ptr(), 1,
                                                 self.comm.cuda_stream), "sr_signal_wait_timeout")
        self._lib.check(L.sr_signal_set(word.data_ptr(), one.data_ptr(), cur.cuda_stream), "sr_signal_set")
        torch.cuda.synchronize(dev)
        return {"independent": not bool(status.any()), "ms": round((time.perf_counter() - t0) * 1e3, 3)}

    def guard(self, optimiser):
        """The optimiser that consumes this buffer refuses a step whose buckets carry the NaN marker of a timed-out
        wait (optim.FlatAdam.set_guards); check() raises when it did."""
        if self.enabled and hasattr(optimiser, "set_guards"):
            if len(self.buckets) > 16:
                raise RuntimeError("BucketedGradReducer.guard: sr_adam_flat_guarded takes at most 16 guard positions "
                                   "(%d buckets; SR_GRAD_BUCKETS <= 16)" % len(self.buckets))
            self.guarded.append(optimiser.set_guards([b["lo"] for b in self.buckets]))
        return optimiser

    # ---- one backward ---------------------------------------------------------------------------------------
    def _claim(self, i):
        if not getattr(self, "active", False) or i in self.claimed:
            return None
        self.claimed.add(i)
        v = self.views[i]
        return v.view_as(v)              # a fresh tensor object over the slot: AccumulateGrad can take it as it is

    def begin(self):
        for p in self.params:
            p.grad = None                # autograd then ASSIGNS: no zero fill, no accumulate kernel
        self.claimed = set()
        self.pending = [len(b["members"]) for b in self.buckets]
        self.done = [False] * len(self.buckets)
        self.next_issue = 0
        self.works = []
        self.log = []
        self.active = True

    def _now(self):
        import time

        return time.perf_counter()

    def _capturing(self):
        return self.is_cuda and torch.cuda.is_current_stream_capturing()

    def _arrived(self, i):
        if not self.active:
            return
        b = self.bucket_of[i]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._flush(b)

    def _flush(self, b):
        members = self.buckets[b]["members"]
        have = [(self.views[i], self.params[i].grad) for i in members if self.params[i].grad is not None]
        # gradients their producer wrote in place (see __init__): nothing to copy
        have = [(v, g) for v, g in have if not (g.data_ptr() == v.data_ptr() and g.shape == v.shape and g.is_contiguous())]
        miss = [self.views[i] for i in members if self.params[i].grad is None]
        with torch.no_grad():
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            if miss:
                torch._foreach_zero_(miss)          # a parameter this phase did not reach
        for i in members:
            self.params[i].grad = self.views[i]     # what the optimiser (graph) was built on
        self.done[b] = True
        self.log.append(("flush", b, self._now()))
        if not self.enabled:
            return
        if self.is_cuda:
            stream = torch.cuda.current_stream(self.flat.device)      # the producing stream (autograd thread: the op's)
            if self._capturing():
                L = self._lib.lib()
                set_word = L.sr_signal_set_host if self.release == "host" else L.sr_signal_set
                self._lib.check(set_word(self.counters.data_ptr() + 4 * b, self.epoch_dev.data_ptr(), stream.cuda_stream),
                                "sr_signal_set")
                return                               # issued by issue_all() after every replay
            ev = torch.cuda.Event()
            ev.record(stream)
            self.eager_events[b] = ev
        self._issue_ready()

    def _issue_ready(self):
        while self.next_issue < len(self.buckets) and self.done[self.next_issue]:
            self._issue(self.next_issue)
            self.next_issue += 1

    def _issue(self, b, replay=False):
        bk = self.buckets[b]
        piece = self.flat[bk["lo"]:bk["hi"]]
        self.log.append(("issue", b, self._now()))
        if self.is_cuda:
            if replay and self.release == "host":
                self._host_wait(b)                   # returns when the replay has passed the bucket's signal node
            elif replay:
                # (on expiry: NaN into the bucket's first element, see the class docstring)
                self._lib.check(self._lib.lib().sr_signal_wait_poison(
                    self.counters.data_ptr() + 4 * b, self.epoch & 0xFFFFFFFF, self.timeout_us,
                    self.status.data_ptr() + 4 * b, b + 1, self.flat.data_ptr() + 4 * bk["lo"],
                    self.comm.cuda_stream), "sr_signal_wait_poison")
            else:
                self.comm.wait_event(self.eager_events[b])
            with torch.cuda.stream(self.comm):
                if self.single is not None:
                    self.single()
                else:
                    dist.all_reduce(piece, op=dist.ReduceOp.SUM)
                    piece.div_(self.world)
        elif self.single is not None:
            self.single()
        else:
            self.works.append((dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True), piece))

    def _host_wait(self, b):
        """Host-released mode: poll the bucket's pinned word until the replay's signal node has stored this epoch.
        A signal that does not come within SR_SIGNAL_TIMEOUT_S is handled like the device mode's expired wait (ADVICE r5:
        raising HERE would leave the peers inside the collective until their RCCL timeout): the bucket's status word is
        set, NaN goes into the bucket's guard position on the communication stream, and the collective is STILL queued
        — it spreads the marker, every rank's guarded optimiser refuses the step and check() raises on every rank."""
        import time

        want = self.epoch & 0xFFFFFFFF
        t0 = time.perf_counter()
        spins = 0
        while ((int(self._words[b]) - want) & 0xFFFFFFFF) >= 0x80000000:
            spins += 1
            if spins & 0x3FF == 0:
                if time.perf_counter() - t0 > self.timeout_us / 1e6:
                    self.status[b] = b + 1
                    lo = self.buckets[b]["lo"]
                    with torch.cuda.stream(self.comm):
                        self.flat[lo:lo + 1].fill_(float("nan"))
                    _log.error("BucketedGradReducer (host release): the signal of bucket %d (of %d) was not published "
                               "within %.0f s at epoch %d on rank %d; bucket poisoned, collective still issued",
                               b, len(self.buckets), self.timeout_us / 1e6, self.epoch, get_rank())
                    return
                time.sleep(0)

    def finish(self):
        """End of the backward: buckets with parameters the phase did not reach are completed (zeros) and signalled."""
        for b in range(len(self.buckets)):
            if not self.done[b]:
                self._flush(b)
        self.active = False

    # ---- around a replay / before the optimiser ----------------------------------------------------------------
    def check(self):
        """Raises when a device-side wait gave up (its bucket's signal node never ran within SR_SIGNAL_TIMEOUT_S), or
        when a guarded optimiser refused a step because a bucket carried the NaN marker of such a wait — on this rank
        or, through the all-reduce, on a peer."""
        if self.enabled and any(bool(w.any()) for w in self.guarded):
            for w in self.guarded:
                w.zero_()
            lost = [int(x) - 1 for x in self.status.tolist() if x] if self.is_cuda else []
            if self.is_cuda:
                self.status.zero_()
            raise RuntimeError("BucketedGradReducer: the optimiser step after epoch %d was REFUSED on rank %d: a gradient "
                               "bucket carried NaN at its guard position (%s) — parameters and moments are unchanged" % (
                                   self.epoch if self.is_cuda else -1, get_rank(),
                                   "this rank's wait for bucket(s) %s timed out" % lost if lost else
                                   "a peer rank's bucket wait timed out, or the gradients themselves are NaN"))
        if self.enabled and self.is_cuda and bool(self.status.any()):
            stuck = [int(x) - 1 for x in self.status.tolist() if x]
            self.status.zero_()
            raise RuntimeError("BucketedGradReducer: the signal of bucket(s) %s (of %d) was not published within %.0f s "
                               "at epoch %d on rank %d — the replay that should produce it did not run or did not "
                               "finish; the gradient buffer is NOT reduced" % (
                                   stuck, len(self.buckets), self.timeout_us / 1e6, self.epoch, get_rank()))

    def arm(self):
        """In front of `graph.replay()` of a captured phase: announce the replay (epoch += 1) to its signal nodes."""
        if self.enabled and self.is_cuda:
            self.check()
            self.epoch += 1
            u = self.epoch & 0xFFFFFFFF                 # the device word is the epoch mod 2^32 (int32 bit pattern)
            self.epoch_dev.fill_(u - (1 << 32) if u >> 31 else u)

    def issue_all(self, stamps=None):
        """After an arm()ed `graph.replay()` of a captured phase (ONLY then: the waits spin on the device until the
        replay's signal nodes have published the epoch): queue every bucket's wait + collective on the comm stream.
        `stamps` (a list): a timing event is recorded on the comm stream after every bucket's collective."""
        if self.enabled:
            if self.is_cuda:
                self.check()
            for b in range(len(self.buckets)):
                self._issue(b, replay=True)
                if stamps is not None and self.is_cuda:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(self.comm)
                    stamps.append(ev)

    def wait(self, deadline_s=None):
        """The current stream (CPU: the caller) waits for all reductions.  deadline_s (or SR_COMM_WATCHDOG_S): a
        host-side watchdog — the caller polls the communication stream until it drains, raising with the stuck
        bucket (check()) or, at the deadline, with the device words it can still read from pinned memory."""
        if not self.enabled:
            return
        if self.is_cuda:
            self.check()
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm)
            if deadline_s is None and os.environ.get("SR_COMM_WATCHDOG_S"):
                deadline_s = float(os.environ["SR_COMM_WATCHDOG_S"])
            if deadline_s:
                import time

                t0 = time.perf_counter()
                while not self.comm.query():
                    self.check()
                    if time.perf_counter() - t0 > deadline_s:
                        raise RuntimeError("BucketedGradReducer: communication stream still busy %.1f s after the "
                                           "replay of epoch %d on rank %d (%d buckets; a peer rank may not have "
                                           "entered the collective)" % (deadline_s, self.epoch, get_rank(),
                                                                        len(self.buckets)))
                    time.sleep(0.0005)
                self.check()
        else:
            for work, piece in self.works:
                work.wait()
                piece.div_(self.world)
            self.works = []

    def describe(self):
        mb = [round((b["hi"] - b["lo"]) * 4 / 1e6, 1) for b in self.buckets]
        mode = "off" if not self.enabled else ("after-backward (%s)" % self.single.mode if self.single is not None
                                               else "overlapped, %d buckets" % len(self.buckets))
        extra = {}
        if self.enabled and self.is_cuda:
            extra = {"release": self.release, "release_probe": self.release_probe}
        return {"mode": mode, "bucket_MB": mb, **extra}
