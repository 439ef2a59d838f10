"""The training iteration of train.Trainer captured into hipGraphs (torch.cuda.CUDAGraph on ROCm).

Why: at the reference's per-GPU batch (4 images, BASELINE config[2]) one iteration enqueues ~4 000 kernels;
Python + launch overhead (~14 us each, 58 ms) exceeds the GPU time of the kernels, and the launch gaps leave the
matrix cores idle.  Every C-ABI entry point of libstylerenderer_hip.so is allocation-free and never synchronises
the host, so the forward + backward of each phase records into a graph unchanged.

One iteration = up to four phases, each `graph(zero grads -> forward -> backward)`, then — OUTSIDE any graph —
one all-reduce of the phase's flat gradient buffer over RCCL (world > 1), then `graph(Adam step)`:

    D      fake = G(z, mesh) (no grad) ; D(fake | real interleaved) ; logistic loss         train.py:245-268
    R1     every d_reg_every: r1/2 * |grad_x D(real)|^2 * d_reg_every                       train.py:281-289
    G      non-saturating loss through D                                                    train.py:292-333
    path   every g_reg_every: path-length regulariser on batch // path_batch_shrink         train.py:335-354
    EMA    g_ema <- decay * g_ema + (1 - decay) * g  (two multi-tensor launches, eager)     train.py:358

Keeping the collectives out of the graphs is deliberate: the data path is then exactly "replay, all-reduce one
contiguous 125 MB / 115 MB buffer, replay" (no NCCL kernels inside captured work, nothing that depends on RCCL's
capture support), at the cost of not overlapping the reduction with the backward (ring bound ~1.4 ms of a ~20 ms
phase on 8 x xGMI).  The eager Trainer keeps the overlapped DDP path.

What has to be static for capture, and how:
  * inputs live in fixed buffers (`real`, two meshes) refreshed by copies before the replays;
  * latents and per-layer noise are drawn INSIDE the graphs (Philox state is graph-safe); style mixing
    (reference train.py:140-144, model.py:160-171) always draws two latents and takes the crossover index from a
    device scalar — crossover == n_latent reproduces "no mixing" exactly;
  * each phase starts with `.grad = None` (autograd then ASSIGNS gradients: no zero fill, no accumulate kernel per
    parameter) and ends with one multi-tensor copy of the phase's gradients into views of a flat buffer per
    network — the buffer the all-reduce and the captured Adam step read;
  * Adam is one captured launch over flat parameter / gradient / moment buffers (optim.FlatAdam, csrc sr_adam_flat);
    the lazy-regularisation cadence is host control flow between replays;
  * loss scalars land in a fixed tensor; nothing is read back unless `log=True`.
"""
import torch

from . import distributed as sr_dist
from .optim import FlatAdam, flat_layout, flat_views
from .train import (Trainer, accumulate, d_logistic_loss, d_r1_loss, g_nonsaturating_loss, g_path_regularize,
                    requires_grad)

LOSS_SLOTS = ("d", "real_score", "fake_score", "r1", "g", "path", "path_length", "mean_path")


class GraphedTrainer(Trainer):
    def __init__(self, *args, batch=4, mesh_vertices=None, capture=True, **kw):
        super().__init__(*args, wrap_ddp=False, **kw)
        if self.device.type != "cuda":
            raise RuntimeError("GraphedTrainer needs a GPU (hipGraph capture); use train.Trainer on CPU")
        if self.args["augment"]:
            raise RuntimeError("GraphedTrainer: the ADA branch reads statistics on the host; use train.Trainer")
        a = self.args
        dev = self.device
        self.batch = batch
        self.capture = capture
        g, d = self.generator, self.discriminator
        # plain modules: gradients are reduced explicitly between the replays
        self.g_ddp, self.d_ddp = g, d
        self.world = sr_dist.get_world_size()
        if self.world > 1:
            for p in list(g.parameters()) + list(d.parameters()) + list(g.buffers()) + list(d.buffers()):
                torch.distributed.broadcast(p.data, 0)
        self.g_params = [p for n, p in g.named_parameters() if n not in self.frozen]
        self.d_params = list(d.parameters())
        self.flat_g, self.views_g = self._flatten_grads(self.g_params, self.world)
        self.flat_d, self.views_d = self._flatten_grads(self.d_params, self.world)
        for p, v in zip(self.g_params + self.d_params, self.views_g + self.views_d):
            p.grad = v
        g_ratio = a["g_reg_every"] / (a["g_reg_every"] + 1)
        d_ratio = a["d_reg_every"] / (a["d_reg_every"] + 1)
        # one-launch Adam over flat parameter / gradient / moment buffers (optim.FlatAdam)
        self.g_optim = FlatAdam(self.g_params, self.flat_g, lr=a["lr"] * g_ratio, betas=(0 ** g_ratio, 0.99 ** g_ratio))
        self.d_optim = FlatAdam(self.d_params, self.flat_d, lr=a["lr"] * d_ratio, betas=(0 ** d_ratio, 0.99 ** d_ratio))
        size = a["size"]
        self.s_real = torch.zeros(batch, 3, size, size, device=dev)
        self.s_inject = {k: torch.zeros((), dtype=torch.int64, device=dev) for k in ("d", "g", "path")}
        self.s_loss = {k: torch.zeros((), device=dev) for k in LOSS_SLOTS}
        # gradient averaging between the replays: all-reduce or in-place reduce-scatter + all-gather, whichever
        # this node's RCCL runs faster on the real buffers (distributed.FlatGradReducer)
        self.reduce_g = sr_dist.FlatGradReducer(self.flat_g, self.world)
        self.reduce_d = sr_dist.FlatGradReducer(self.flat_d, self.world)
        self.s_mesh = None
        if self.use_mesh:
            if mesh_vertices is None:
                raise ValueError("GraphedTrainer(use_mesh=True) needs mesh_vertices (and set_topology(tri))")
            self.s_mesh = {k: (torch.zeros(batch, mesh_vertices, 3, device=dev),
                               torch.zeros(batch, mesh_vertices, 3, device=dev)) for k in ("d", "g")}
            self.tri = None
        self.graphs = {}

    # ---- static state -----------------------------------------------------------------------------
    @staticmethod
    def _flatten_grads(params, world=1):
        offs, total = flat_layout(params, world)               # 256-byte aligned slots, shared with FlatAdam
        flat = torch.zeros(total, device=params[0].device, dtype=params[0].dtype)
        return flat, flat_views(flat, params, offs)

    @staticmethod
    def _clear(params):
        for p in params:
            p.grad = None

    @staticmethod
    def _collect(params, views):
        """Phase gradients -> flat buffer (one multi-tensor copy); parameters a phase did not reach get zeros.
        Afterwards `.grad` IS the flat view, which is what the optimiser graph was captured on."""
        have = [(v, p.grad) for p, v in zip(params, views) if p.grad is not None]
        miss = [v for p, v in zip(params, views) if p.grad is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if miss:
            torch._foreach_zero_(miss)
        for p, v in zip(params, views):
            p.grad = v

    def set_topology(self, tri):
        self.tri = tri.contiguous()

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

    # ---- the four phases (pure device work: these bodies are what the graphs record) ---------------
    def _phase_d(self):
        g, d = self.generator, self.discriminator
        requires_grad(d, True)
        self._clear(self.d_params)
        with torch.no_grad():
            fake, _, _ = self._generate(g, self._latents(self.batch), self._mesh_tuple("d"),
                                        inject_index=self.s_inject["d"])
        both = torch.stack([fake, self.s_real], 1).reshape(2 * self.batch, *self.s_real.shape[1:])
        pred = d(both)
        fake_pred, real_pred = pred[0::2], pred[1::2]
        loss = d_logistic_loss(real_pred, fake_pred)
        loss.backward()
        self._collect(self.d_params, self.views_d)
        self.s_loss["d"].copy_(loss.detach())
        self.s_loss["real_score"].copy_(real_pred.detach().mean())
        self.s_loss["fake_score"].copy_(fake_pred.detach().mean())

    def _phase_r1(self):
        d = self.discriminator
        requires_grad(d, True)
        self._clear(self.d_params)
        real = self.s_real.detach().clone().requires_grad_(True)
        pred = d(real)
        r1 = d_r1_loss(pred, real)
        (self.args["r1"] / 2 * r1 * self.args["d_reg_every"] + 0 * pred[0]).backward()
        self._collect(self.d_params, self.views_d)
        self.s_loss["r1"].copy_(r1.detach())

    def _phase_g(self):
        g, d = self.generator, self.discriminator
        requires_grad(d, False)
        self._clear(self.g_params)
        fake, _, _ = self._generate(g, self._latents(self.batch), self._mesh_tuple("g"),
                                    inject_index=self.s_inject["g"])
        loss = g_nonsaturating_loss(d(fake))
        loss.backward()
        self._collect(self.g_params, self.views_g)
        self.s_loss["g"].copy_(loss.detach())

    def _phase_path(self):
        a = self.args
        g = self.generator
        requires_grad(self.discriminator, False)
        self._clear(self.g_params)
        pb = max(1, self.batch // a["path_batch_shrink"]) if a["path_batch_shrink"] else self.batch
        fake, latents, normals = self._generate(g, self._latents(pb), self._mesh_tuple("g", pb), return_latents=True,
                                                return_normals=True, inject_index=self.s_inject["path"])
        targets = [latents] + (list(normals) if normals else [])
        path_loss, path_mean, path_lengths = g_path_regularize(fake, targets, self.mean_path_length)
        weighted = a["path_regularize"] * a["g_reg_every"] * path_loss
        if a["path_batch_shrink"]:
            weighted = weighted + 0 * fake[0, 0, 0, 0]
        weighted.backward()
        self._collect(self.g_params, self.views_g)
        self.mean_path_length.copy_(path_mean)
        self.s_loss["path"].copy_(path_loss.detach())
        self.s_loss["path_length"].copy_(path_lengths.detach().mean())
        self.s_loss["mean_path"].copy_(path_mean)

    # ---- capture -------------------------------------------------------------------------------------
    def _bodies(self):
        return {"d": self._phase_d, "r1": self._phase_r1, "g": self._phase_g, "path": self._phase_path,
                "d_opt": self.d_optim.step, "g_opt": self.g_optim.step}

    def build_graphs(self, warmup=3):
        """Eager warm-up of every phase on a side stream (lazy initialisation, Adam state, incidence caches),
        then one capture per phase.  The warm-up iterations are real optimisation steps."""
        bodies = self._bodies()
        for k in self.s_inject:
            self._draw_inject(k)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        reduce_after = {"d": self.flat_d, "r1": self.flat_d, "g": self.flat_g, "path": self.flat_g}
        with torch.cuda.stream(side):
            for _ in range(warmup):
                for name in ("d", "d_opt", "r1", "d_opt", "g", "g_opt", "path", "g_opt"):
                    bodies[name]()
                    if name in reduce_after:
                        self._reduce(reduce_after[name])       # replicas stay in step during the warm-up too
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if not self.capture:
            return
        # thread_local: only THIS thread's calls are policed during capture.  Under the default (global) mode the
        # RCCL watchdog thread's routine hipEventQuery on an earlier collective (the warm-up reductions, a DDP leg
        # that ran before) aborts the process with "operation not permitted when stream is capturing".
        for name in ("d", "r1", "g", "path", "d_opt", "g_opt"):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                bodies[name]()
            self.graphs[name] = graph
        torch.cuda.synchronize()

    def _run(self, name):
        if self.capture:
            self.graphs[name].replay()
        else:
            self._bodies()[name]()

    def _reduce(self, flat):
        if self.world > 1:
            (self.reduce_g if flat is self.flat_g else self.reduce_d)()

    # ---- one iteration -------------------------------------------------------------------------------
    def step(self, real_img, mesh=None, faces=None, log=True):
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
        self._reduce(self.flat_d)
        self._run("d_opt")
        ran = ["d", "real_score", "fake_score", "g"]
        if i % a["d_reg_every"] == 0:
            self._run("r1")
            self._reduce(self.flat_d)
            self._run("d_opt")
            ran.append("r1")
        self._run("g")
        self._reduce(self.flat_g)
        self._run("g_opt")
        if i % a["g_reg_every"] == 0:
            self._run("path")
            self._reduce(self.flat_g)
            self._run("g_opt")
            ran += ["path", "path_length", "mean_path"]
        accumulate(self.g_ema, self.generator, self.accum)
        self.iteration += 1
        return sr_dist.reduce_scalars({k: self.s_loss[k] for k in ran}, to_host=log)

    def _load_inputs(self, real_img, mesh, faces):
        self.s_real.copy_(real_img)
        if self.use_mesh:
            if self.tri is None:
                self.set_topology(faces.tri if faces is not None else mesh[2])
            for key in ("d", "g"):
                v, nrm, _ = faces.sample(self.batch) if faces is not None else mesh
                self.s_mesh[key][0].copy_(v)
                self.s_mesh[key][1].copy_(nrm)
