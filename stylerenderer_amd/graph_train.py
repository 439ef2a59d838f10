"""The training iteration of train.Trainer captured into hipGraphs (torch.cuda.CUDAGraph on ROCm).

Why: at the reference's per-GPU batch (4 images, BASELINE config[2]) one iteration enqueues ~4 000 kernels;
Python + launch overhead (~14 us each, 58 ms) exceeds the GPU time of the kernels, and the launch gaps leave the
matrix cores idle.  Every C-ABI entry point of libstylerenderer_hip.so is allocation-free and never synchronises
the host, so the forward + backward of each phase records into a graph unchanged.

One iteration = up to four phases, each `graph(forward -> backward)`; the phase's flat gradient buffer is averaged over
the ranks by RCCL collectives that run OUTSIDE the graphs but DURING the replay (below); then `graph(Adam step)`:

    D      fake = G(z, mesh) (no grad) ; D(fake | real interleaved) ; logistic loss         train.py:245-268
    R1     every d_reg_every: r1/2 * |grad_x D(real)|^2 * d_reg_every                       train.py:281-289
    G      non-saturating loss through D                                                    train.py:292-333
    path   every g_reg_every: path-length regulariser on batch // path_batch_shrink         train.py:335-354
    EMA    g_ema <- decay * g_ema + (1 - decay) * g  (two multi-tensor launches, eager)     train.py:358

Gradient reduction overlapped with the backward (reference distributed.py:98-105, train.py:335-352; north_star: "RCCL
all-reduce over xGMI overlapped with the StyleGAN2 path-length-regulariser backward").  No RCCL call is captured:
the flat gradient buffer of each network is laid out in gradient ARRIVAL order and cut into SR_GRAD_BUCKETS (4)
contiguous buckets; the captured backward holds, after the last gradient of each bucket, one multi-tensor copy into the
bucket's flat views and a SIGNAL NODE (`sr_signal_set`: a one-lane kernel that publishes the replay's epoch — a device
scalar the host bumps in front of every replay — into the bucket's device word; include/stylerenderer_amd.h —
event-record nodes are refused by the HIP runtime torch bundles).  Right after `graph.replay()` the host queues, for
every bucket, `sr_signal_wait_timeout(word_k, epoch)` + `all_reduce(flat[lo_k:hi_k])` on a communication stream: bucket k is on the xGMI links while the replay
is still producing bucket k+1; the optimiser graph waits for the communication stream
(distributed.BucketedGradReducer).  SR_GRAD_OVERLAP=0: one bucket, reduced after the backward by the autotuned
all-reduce / reduce-scatter + all-gather of distributed.FlatGradReducer (round 2's mode).  With `capture=False` the
same phases, hooks and collectives run eagerly — on CPU tensors over gloo too, which is how the world-size-2 tests
exercise this trainer without a GPU.

What has to be static for capture, and how:
  * inputs live in fixed buffers (`real`, two meshes) refreshed by copies before the replays;
  * latents and per-layer noise are drawn INSIDE the graphs (Philox state is graph-safe); style mixing
    (reference train.py:140-144, model.py:160-171) always draws two latents and takes the crossover index from a
    device scalar — crossover == n_latent reproduces "no mixing" exactly;
  * each phase starts with `.grad = None` (autograd then ASSIGNS gradients: no zero fill, no accumulate kernel per
    parameter); a completed bucket is moved into views of a flat buffer per network by one multi-tensor copy — the
    buffer the collectives and the captured Adam step read;
  * Adam is one captured launch over flat parameter / gradient / moment buffers (optim.FlatAdam, csrc sr_adam_flat);
    the lazy-regularisation cadence is host control flow between replays;
  * loss scalars land in a fixed tensor; nothing is read back unless `log=True`.
"""
import os

import torch

from . import distributed as sr_dist
from . import graphs
from .optim import FlatAdam, flat_layout, flat_views
from .train import (Trainer, accumulate, d_fake_real, d_logistic_loss, d_r1_loss, g_nonsaturating_loss, g_path_regularize,
                    requires_grad)

LOSS_SLOTS = ("d", "real_score", "fake_score", "r1", "g", "path", "path_length", "mean_path")


class GraphedTrainer(Trainer):
    def __init__(self, *args, batch=4, mesh_vertices=None, capture=True, n_buckets=None, force_collectives=False, **kw):
        super().__init__(*args, wrap_ddp=False, **kw)
        on_gpu = self.device.type == "cuda"
        if capture and not on_gpu:
            raise RuntimeError("GraphedTrainer(capture=True) needs a GPU (hipGraph capture); CPU tensors run the same "
                               "phases eagerly with capture=False")
        if self.args["augment"]:
            raise RuntimeError("GraphedTrainer: the ADA branch reads statistics on the host; use train.Trainer")
        a = self.args
        dev = self.device
        self.batch = batch
        self.capture = capture
        g, d = self.generator, self.discriminator
        # plain modules: gradients are reduced explicitly, bucket by bucket, while the phase runs
        self.g_ddp, self.d_ddp = g, d
        self.world = sr_dist.get_world_size()
        if self.world > 1:
            for p in list(g.parameters()) + list(d.parameters()) + list(g.buffers()) + list(d.buffers()):
                torch.distributed.broadcast(p.data, 0)
        self.g_params = [p for n, p in g.named_parameters() if n not in self.frozen]
        self.d_params = list(d.parameters())
        size = a["size"]
        self.s_mesh = None
        self.tri = None
        if self.use_mesh:
            if mesh_vertices is None:
                raise ValueError("GraphedTrainer(use_mesh=True) needs mesh_vertices (and set_topology(tri))")
            self.s_mesh = {k: (torch.zeros(batch, mesh_vertices, 3, device=dev),
                               torch.zeros(batch, mesh_vertices, 3, device=dev)) for k in ("d", "g")}
        overlap = os.environ.get("SR_GRAD_OVERLAP", "1") != "0"
        if n_buckets is None:
            n_buckets = int(os.environ.get("SR_GRAD_BUCKETS", "4")) if overlap else 1
        # flat gradient buffers in gradient-arrival order (one cheap probe backward per network, rank 0's order)
        order_g = sr_dist.arrival_order(self.g_params, self._probe_g)
        order_d = sr_dist.arrival_order(self.d_params, self._probe_d)
        self.flat_g, self.views_g, offs_g = self._flatten_grads(self.g_params, self.world, order_g)
        self.flat_d, self.views_d, offs_d = self._flatten_grads(self.d_params, self.world, order_d)
        for p, v in zip(self.g_params + self.d_params, self.views_g + self.views_d):
            p.grad = v
        g_ratio = a["g_reg_every"] / (a["g_reg_every"] + 1)
        d_ratio = a["d_reg_every"] / (a["d_reg_every"] + 1)
        # one-launch Adam over flat parameter / gradient / moment buffers (optim.FlatAdam)
        self.g_optim = FlatAdam(self.g_params, self.flat_g, lr=a["lr"] * g_ratio, betas=(0 ** g_ratio, 0.99 ** g_ratio),
                                offs=offs_g)
        self.d_optim = FlatAdam(self.d_params, self.flat_d, lr=a["lr"] * d_ratio, betas=(0 ** d_ratio, 0.99 ** d_ratio),
                                offs=offs_d)
        self.s_real = torch.zeros(batch, 3, size, size, device=dev)
        self.s_inject = {k: torch.zeros((), dtype=torch.int64, device=dev) for k in ("d", "g", "path")}
        self.s_loss = {k: torch.zeros((), device=dev) for k in LOSS_SLOTS}
        # gradient averaging overlapped with the backward of every phase (distributed.BucketedGradReducer)
        self.reduce_g = sr_dist.BucketedGradReducer(self.g_params, self.views_g, offs_g, self.flat_g, self.world,
                                                    n_buckets, force=force_collectives)
        self.reduce_d = sr_dist.BucketedGradReducer(self.d_params, self.views_d, offs_d, self.flat_d, self.world,
                                                    n_buckets, force=force_collectives)
        # a bucket wait that times out marks its bucket; the Adam step of the buffer then refuses the update on every rank
        self.reduce_g.guard(self.g_optim)
        self.reduce_d.guard(self.d_optim)
        # the three networks are rewritten by REPLAYED graphs from here on (Adam, EMA): replays do not bump tensor versions,
        # so version-keyed frozen caches would go stale — op.weight_prep.freeze_prepared_weights refuses these modules
        # (sample / invert from copy.deepcopy(trainer.g_ema))
        from .op.weight_prep import GRAPH_WRITTEN

        for net in (self.generator, self.discriminator, self.g_ema):
            GRAPH_WRITTEN.add(net)
        self.graphs = {}
        # a mesh SOURCE handed to step(faces=...) is sampled inside the D and G phases (reference train.py:248-251,
        # 303-306 draws a fresh batch of meshes in front of each of the two generator passes): ~40 small launches per
        # iteration that would otherwise run eagerly between two replays, where the host cannot run ahead of the GPU
        self.face_source = None

    # ---- gradient-arrival probes (layout of the flat buffers) -------------------------------------------
    def _probe_g(self):
        """One batch-1 forward + backward of the generator on a dummy mesh (a single degenerate triangle: the
        rasterised maps are empty, every layer still takes part in the backward)."""
        g = self.generator
        z = [torch.zeros(1, self.args["latent"], device=self.device)]
        if self.use_mesh:
            nv = self.s_mesh["g"][0].shape[1]
            mesh = (torch.zeros(1, nv, 3, device=self.device), torch.zeros(1, nv, 3, device=self.device),
                    torch.zeros(1, 3, dtype=torch.int64, device=self.device))
            img = g(z, mesh)[0]
        else:
            img = g(z)[0]
        img.sum().backward()

    def _probe_d(self):
        d = self.discriminator
        x = torch.zeros(d.stddev_group, 3, self.args["size"], self.args["size"], device=self.device)
        d(x).sum().backward()

    # ---- static state -----------------------------------------------------------------------------
    @staticmethod
    def _flatten_grads(params, world=1, order=None):
        offs, total = flat_layout(params, world, order)        # 256-byte aligned slots, shared with FlatAdam
        flat = torch.zeros(total, device=params[0].device, dtype=params[0].dtype)
        return flat, flat_views(flat, params, offs), offs

    def set_topology(self, tri):
        """The shared triangle list (made contiguous ONCE: the per-topology incidence lists of the rasterizer gradient
        and of the vertex normals are cached by its address, and a cache miss inside a capture would run torch.sort /
        bincount, which synchronise the host and abort the capture).  Both lists are built here."""
        self.tri = tri.contiguous()
        if self.tri.is_cuda and self.s_mesh is not None:
            from . import utils_3d
            from .op.rasterize import incidence

            nv = self.s_mesh["g"][0].shape[1]
            incidence(self.tri, nv)
            utils_3d.incidence_lists(self.tri, nv)

    def _mesh_tuple(self, key, n=None):
        if not self.use_mesh:
            return None
        v, nrm = self.s_mesh[key]
        if n is not None:
            return v[:n].detach().requires_grad_(True), nrm[:n].detach().requires_grad_(True), self.tri
        return v, nrm, self.tri

    def _latents(self, n):
        z = torch.randn(2, n, self.args["latent"], device=self.device)
        return [z[0], z[1]]

    def _draw_inject(self, key):
        """Host side of mixing_noise (reference train.py:140-144): with probability `mixing` a crossover in
        [1, n_latent - 2] (model.py:168), else n_latent (= a single latent everywhere)."""
        n_latent = self.generator.n_latent
        k = int(self.np_rng.randint(n_latent - 2)) + 1 if (self.args["mixing"] > 0 and
                                                           self.np_rng.rand() < self.args["mixing"]) else n_latent
        self.s_inject[key].fill_(k)

    def _sample_mesh(self, key):
        if self.face_source is None or not self.use_mesh:
            return
        with torch.no_grad():
            v, nrm, _ = self.face_source.sample(self.batch)
            self.s_mesh[key][0].copy_(v)
            self.s_mesh[key][1].copy_(nrm)

    def _phase_ema(self):
        accumulate(self.g_ema, self.generator, self.accum)

    # ---- the four phases (pure device work: these bodies are what the graphs record) ---------------
    def _phase_d(self):
        g, d = self.generator, self.discriminator
        requires_grad(d, True)
        self.reduce_d.begin()
        self._sample_mesh("d")
        with torch.no_grad():
            fake, _, _ = self._generate(g, self._latents(self.batch), self._mesh_tuple("d"),
                                        inject_index=self.s_inject["d"])
        fake_pred, real_pred = d_fake_real(d, d, fake, self.s_real)      # one interleaved pass when batch % 4 == 0
        loss = d_logistic_loss(real_pred, fake_pred)
        loss.backward()
        self.reduce_d.finish()
        self.s_loss["d"].copy_(loss.detach())
        self.s_loss["real_score"].copy_(real_pred.detach().mean())
        self.s_loss["fake_score"].copy_(fake_pred.detach().mean())

    def _phase_r1(self):
        d = self.discriminator
        requires_grad(d, True)
        self.reduce_d.begin()
        real = self.s_real.detach().clone().requires_grad_(True)
        pred = d(real)
        r1 = d_r1_loss(pred, real)
        (self.args["r1"] / 2 * r1 * self.args["d_reg_every"] + 0 * pred[0]).backward()
        self.reduce_d.finish()
        self.s_loss["r1"].copy_(r1.detach())

    def _phase_g(self):
        g, d = self.generator, self.discriminator
        requires_grad(d, False)
        self.reduce_g.begin()
        self._sample_mesh("g")             # the path phase reuses this batch's first meshes (train.py:337-338)
        fake, _, _ = self._generate(g, self._latents(self.batch), self._mesh_tuple("g"),
                                    inject_index=self.s_inject["g"])
        loss = g_nonsaturating_loss(d(fake))
        loss.backward()
        self.reduce_g.finish()
        self.s_loss["g"].copy_(loss.detach())

    def _phase_path(self):
        a = self.args
        g = self.generator
        requires_grad(self.discriminator, False)
        self.reduce_g.begin()
        pb = max(1, self.batch // a["path_batch_shrink"]) if a["path_batch_shrink"] else self.batch
        fake, latents, normals = self._generate(g, self._latents(pb), self._mesh_tuple("g", pb), return_latents=True,
                                                return_normals=True, inject_index=self.s_inject["path"])
        targets = [latents] + (list(normals) if normals else [])
        path_loss, path_mean, path_lengths = g_path_regularize(fake, targets, self.mean_path_length)
        weighted = a["path_regularize"] * a["g_reg_every"] * path_loss
        if a["path_batch_shrink"]:
            weighted = weighted + 0 * fake[0, 0, 0, 0]
        weighted.backward()
        self.reduce_g.finish()
        self.mean_path_length.copy_(path_mean)
        self.s_loss["path"].copy_(path_loss.detach())
        self.s_loss["path_length"].copy_(path_lengths.detach().mean())
        self.s_loss["mean_path"].copy_(path_mean)

    # ---- capture -------------------------------------------------------------------------------------
    def _bodies(self):
        return {"d": self._phase_d, "r1": self._phase_r1, "g": self._phase_g, "path": self._phase_path,
                "d_opt": self.d_optim.step, "g_opt": self.g_optim.step, "ema": self._phase_ema}

    PHASE_REDUCER = {"d": "reduce_d", "r1": "reduce_d", "g": "reduce_g", "path": "reduce_g"}

    def _snapshot(self):
        """Everything the warm-up iterations change: parameters, Adam moments / step, the path-length EMA and the
        random streams.  Restored before capture, so training starts (or resumes) from exactly the loaded state and
        the lazy-regularisation cadence is not advanced by untracked steps.  (Module buffers — noise maps, FIR taps —
        are constants no phase writes, and the EMA copy is only touched by step(): neither needs a snapshot.)"""
        snap = {"tensors": [(t, t.detach().clone()) for t in (
            self.g_optim.flat_p, self.g_optim.m, self.g_optim.v, self.g_optim.step_t, self.d_optim.flat_p,
            self.d_optim.m, self.d_optim.v, self.d_optim.step_t, self.mean_path_length)],
            "np": self.np_rng.get_state(), "torch": torch.get_rng_state()}
        if self.device.type == "cuda":
            snap["cuda"] = torch.cuda.get_rng_state(self.device)
        return snap

    def _restore(self, snap):
        with torch.no_grad():
            for t, saved in snap["tensors"]:
                t.copy_(saved)
        self.np_rng.set_state(snap["np"])
        torch.set_rng_state(snap["torch"])
        if "cuda" in snap:
            torch.cuda.set_rng_state(snap["cuda"], self.device)

    def _eager_phase(self, name):
        """A phase body run eagerly: the hooks issue the bucket collectives while the backward runs."""
        self._bodies()[name]()
        if name in self.PHASE_REDUCER:
            getattr(self, self.PHASE_REDUCER[name]).wait()

    def build_graphs(self, warmup=None):
        """Eager warm-up of every phase on a side stream (lazy initialisation, incidence / tap caches, RCCL
        communicators), state restored, then one capture per phase.  SR_GRAPH_WARMUP (default 3) = eager passes; the
        counter-collection probes use 1 (every dispatch costs milliseconds under rocprofv3 --pmc)."""
        if not self.capture:
            return
        if warmup is None:
            warmup = max(1, int(os.environ.get("SR_GRAPH_WARMUP", "3")))
        snap = self._snapshot()
        for k in self.s_inject:
            self._draw_inject(k)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                for name in ("d", "d_opt", "r1", "d_opt", "g", "g_opt", "path", "g_opt"):
                    self._eager_phase(name)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        bodies = self._bodies()
        for name in ("d", "r1", "g", "path", "d_opt", "g_opt", "ema"):
            self.graphs[name] = graphs.capture(bodies[name])     # memset nodes repaired: graphs.py
        torch.cuda.synchronize()

    def _run(self, name):
        """One phase (or optimiser step).  Captured: announce the replay's epoch, replay, then queue every bucket's
        wait-for-signal kernel (sr_signal_wait_timeout) + collective on the communication stream (each starts as soon
        as the replay passes the bucket's signal node, sr_signal_set); the current stream — hence the optimiser graph
        replayed next — waits for the communication stream."""
        if not self.capture:
            return self._eager_phase(name)
        red = getattr(self, self.PHASE_REDUCER[name]) if name in self.PHASE_REDUCER else None
        if red is not None:
            red.arm()
        self.graphs[name].replay()
        if red is not None:
            red.issue_all()
            red.wait()

    def time_phase(self, name, reps=3):
        """Milliseconds per `_run(name)` (replay + the phase's gradient collectives, like an iteration issues them),
        averaged over `reps` after one untimed run.  Every rank must call it (the collectives are part of the phase)."""
        self._run(name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self._run(name)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def measure_overlap(self, name="path"):
        """Replays phase `name` once with timing events: returns the replay's duration and, per bucket, when its
        reduction FINISHED relative to the end of the replay (ms; negative = while the backward was still running).
        Leaves the flat gradient buffer reduced like a normal `_run(name)` (no optimiser step follows)."""
        red = getattr(self, self.PHASE_REDUCER[name])
        if not (self.capture and red.enabled and red.is_cuda):
            return None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        red.arm()
        self.graphs[name].replay()
        e1.record()
        stamps = []
        red.issue_all(stamps)
        red.wait()
        torch.cuda.synchronize()
        return {"phase": name, "replay_ms": round(e0.elapsed_time(e1), 3),
                "bucket_done_ms_after_replay_end": [round(e1.elapsed_time(ev), 3) for ev in stamps],
                **red.describe()}

    # ---- one iteration -------------------------------------------------------------------------------
    def step(self, real_img, mesh=None, faces=None, log=True):
        if faces is not None and self.use_mesh and self.face_source is None and not self.graphs:
            self.face_source = faces
        if self.face_source is not None and (mesh is not None or (faces is not None and faces is not self.face_source)):
            raise RuntimeError("GraphedTrainer: the mesh source is part of the captured phases; it cannot change")
        if not self.graphs and self.capture:
            if self.use_mesh and self.tri is None:
                self.set_topology(faces.tri if faces is not None else mesh[2])
            self._load_inputs(real_img, mesh, faces)
            self.build_graphs()
        a = self.args
        i = self.iteration
        self._load_inputs(real_img, mesh, faces)
        for k in self.s_inject:
            self._draw_inject(k)
        self._run("d")
        self._run("d_opt")
        ran = ["d", "real_score", "fake_score", "g"]
        if i % a["d_reg_every"] == 0:
            self._run("r1")
            self._run("d_opt")
            ran.append("r1")
        self._run("g")
        self._run("g_opt")
        if i % a["g_reg_every"] == 0:
            self._run("path")
            self._run("g_opt")
            ran += ["path", "path_length", "mean_path"]
        self._run("ema")
        self.iteration += 1
        return sr_dist.reduce_scalars({k: self.s_loss[k] for k in ran}, to_host=log)

    def _load_inputs(self, real_img, mesh, faces):
        self.s_real.copy_(real_img)
        if self.use_mesh:
            if self.tri is None:
                self.set_topology(faces.tri if faces is not None else mesh[2])
            if self.face_source is not None and mesh is None:
                return                                    # sampled inside the D / G phases (_sample_mesh)
            for key in ("d", "g"):
                v, nrm, _ = faces.sample(self.batch) if faces is not None else mesh
                self.s_mesh[key][0].copy_(v)
                self.s_mesh[key][1].copy_(nrm)
