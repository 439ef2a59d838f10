"""One G/D training iteration with the reference's semantics on the MI355X-native operators,
data parallel over RCCL.  The reference's train.py does not parse; this module
restates the *step* it describes:

  D step           logistic loss on fake / real                       reference train.py:245-268, 105-109
  R1 (lazy, /16)   r1/2 * |grad_x D(real)|^2 * d_reg_every            reference train.py:281-289, 110-114
  G step           non-saturating loss                                reference train.py:292-333, 115-117
  path reg (/4)    (|J^T y| - EMA)^2 on batch // path_batch_shrink    reference train.py:335-354, 118-134
                   (double backward through every custom operator)
  EMA              g_ema <- decay * g_ema + (1 - decay) * g           reference train.py:100-104, 358
  Adam             lazy-regularisation corrected lr / betas           reference train.py:529-536
  logging          ONE packed all-reduce instead of 7 `.item()` round trips (distributed.reduce_scalars)

`python -m stylerenderer_amd.train --size 256 --batch 4 --iter 64` runs it on a synthetic in-memory
dataset (uniform [-1, 1] images, the tensor contract of reference train.py:557-560) and, with
`--mesh`, on GeneratorWithMap with a formula ellipsoid in place of the licensed 3DMM.
"""
import argparse
import math
import time

import numpy as np
import torch
from torch import autograd, optim
from torch.nn import functional as F

from . import distributed as sr_dist
from . import synth
from .model import Discriminator, Generator, GeneratorWithMap


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad_(flag)


def requires_grad_trainable(model, flag, frozen):
    for n, p in model.named_parameters():
        p.requires_grad_(flag and n not in frozen)


def accumulate(model1, model2, decay=0.999):
    """EMA of the parameters (reference train.py:100-104) as two multi-tensor launches instead of two
    launches per parameter (~300 per iteration at 256x256)."""
    par1, par2 = dict(model1.named_parameters()), dict(model2.named_parameters())
    with torch.no_grad():
        dst = [par1[k] for k in par1]
        src = [par2[k].detach() for k in par1]
        torch._foreach_mul_(dst, decay)
        torch._foreach_add_(dst, src, alpha=1 - decay)


def d_logistic_loss(real_pred, fake_pred):
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    (grad_real,) = autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def d_fake_real(d_call, d, fake_img, real_img):
    """D(fake), D(real) (reference train.py:254-255).  When the per-GPU batch is a multiple of the minibatch-stddev
    group, both halves go through ONE interleaved pass: the layer views the batch as [group, batch // group]
    (reference model.py:325-332), so sample b of the 2B batch lands in sub-batch b mod (2B / group) — fake and real
    never share a sub-batch and the statistics equal those of two calls.  Otherwise two calls, like the reference."""
    batch = real_img.shape[0]
    if batch == fake_img.shape[0] and batch % d.stddev_group == 0:
        both = torch.stack([fake_img, real_img], 1).reshape(2 * batch, *real_img.shape[1:])
        pred = d_call(both)
        return pred[0::2], pred[1::2]
    return d_call(fake_img), d_call(real_img)


def g_nonsaturating_loss(fake_pred):
    return F.softplus(-fake_pred).mean()


def g_path_regularize(fake_img, latents, mean_path_length, decay=0.01, lambda_=1., noise=None):
    """Path-length regulariser (reference train.py:118-134).  `latents`: tensor or list of tensors the
    Jacobian is taken against; `lambda_`: scalar or sequence of per-target weights, padded with ones
    like the reference; `noise` (extension, tests): the image-space probe in place of randn_like."""
    if noise is None:
        noise = torch.randn_like(fake_img)
    noise = noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    if not isinstance(latents, (list, tuple)):
        latents = [latents]
    lambda_ = [float(x) for x in np.reshape(lambda_, -1)]
    lambda_ += [1.0] * (len(latents) - len(lambda_))
    grads = autograd.grad(outputs=(fake_img * noise).sum(), inputs=list(latents), create_graph=True,
                          allow_unused=True)
    path_lengths = 0
    for lam, g in zip(lambda_, grads):
        if g is not None:
            flat = g.reshape(g.shape[0], -1)
            path_lengths = path_lengths + torch.sqrt((flat * flat).sum(1)) * lam
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    path_penalty = (path_lengths - path_mean).pow(2).mean()
    return path_penalty, path_mean.detach(), path_lengths


def make_noise(batch, latent_dim, n_noise, device, generator=None):
    if n_noise <= 0:
        return torch.randn(batch, latent_dim, device=device, generator=generator)
    return torch.randn(n_noise, batch, latent_dim, device=device, generator=generator).unbind(0)


def mixing_noise(batch, latent_dim, prob, device, rng=np.random, generator=None):
    if prob > 0 and rng.rand() < prob:
        return list(make_noise(batch, latent_dim, 2, device, generator))
    return [make_noise(batch, latent_dim, 0, device, generator)]


class Trainer:
    """Owns G, D, the EMA copy and both optimisers; `step()` is one reference iteration
    (reference train.py:239-358)."""

    def __init__(self, size=256, latent=512, n_mlp=8, channel_multiplier=2, lr=0.002, r1=10.0,
                 path_regularize=2.0, path_batch_shrink=2, d_reg_every=16, g_reg_every=4, mixing=0.9,
                 use_mesh=False, device="cpu", seed=0, augment=False, augment_p=0.0, ada_target=0.6,
                 ada_length=500 * 1000, wrap_ddp=True):
        self.args = dict(size=size, latent=latent, r1=r1, path_regularize=path_regularize,
                         path_batch_shrink=path_batch_shrink, d_reg_every=d_reg_every,
                         g_reg_every=g_reg_every, mixing=mixing, augment=augment, augment_p=augment_p,
                         ada_target=ada_target, ada_length=ada_length, lr=lr, n_mlp=n_mlp,
                         channel_multiplier=channel_multiplier, seed=seed)
        self.device = torch.device(device)
        self.use_mesh = use_mesh
        torch.manual_seed(seed)                     # identical initial weights on every rank
        cls = GeneratorWithMap if use_mesh else Generator
        self.generator = cls(size, latent, n_mlp, channel_multiplier=channel_multiplier).to(self.device)
        self.discriminator = Discriminator(size, channel_multiplier=channel_multiplier).to(self.device)
        self.g_ema = cls(size, latent, n_mlp, channel_multiplier=channel_multiplier).to(self.device)
        self.g_ema.load_state_dict(self.generator.state_dict())
        self.g_ema.eval()
        # from here on every rank draws its OWN latents / noise / meshes / poses: the reference seeds
        # each process with seed + rank (reference distributed.py:93-95); a shared stream would make the
        # N replicas' fake batches duplicates of each other
        rank = sr_dist.get_rank()
        torch.manual_seed(seed + 1 + rank)
        sr_dist.freeze_unused_tail(self.generator)
        self.frozen = {n for n, p in self.generator.named_parameters() if not p.requires_grad}
        g_ratio = g_reg_every / (g_reg_every + 1)
        d_ratio = d_reg_every / (d_reg_every + 1)
        self.g_optim = optim.Adam([p for p in self.generator.parameters() if p.requires_grad],
                                  lr=lr * g_ratio, betas=(0 ** g_ratio, 0.99 ** g_ratio))
        self.d_optim = optim.Adam(self.discriminator.parameters(), lr=lr * d_ratio,
                                  betas=(0 ** d_ratio, 0.99 ** d_ratio))
        self.g_ddp = sr_dist.construct_ddp(self.generator, self.device) if wrap_ddp else self.generator
        self.d_ddp = sr_dist.construct_ddp(self.discriminator, self.device) if wrap_ddp else self.discriminator
        self.mean_path_length = torch.zeros((), device=self.device)
        self.accum = 0.5 ** (32 / (10 * 1000))
        self.np_rng = np.random.RandomState(seed + 17 * rank)
        self.iteration = 0
        # adaptive discriminator augmentation state (reference train.py:222-225, 269-280)
        self.ada_aug_p = augment_p if augment_p > 0 else 0.0
        self.ada_augment = torch.zeros(2, device=self.device)
        self.r_t_stat = 0.0

    def _generate(self, net, noise, mesh, **kw):
        if self.use_mesh:
            return net(noise, mesh, **kw)
        out = net(noise, **{k: v for k, v in kw.items() if k != "return_normals"})
        return out[0], out[1], None

    def _mesh(self, mesh, faces, batch):
        if not self.use_mesh:
            return None
        return faces.sample(batch) if faces is not None else mesh

    def _augment(self, img):
        if not self.args["augment"]:
            return img
        from .utils_3d import augment

        return augment(img, self.ada_aug_p)

    def step(self, real_img, mesh=None, faces=None, log=True):
        """One iteration.  `faces` (an object with .sample(batch) -> (vert, normals, tri), e.g.
        SyntheticFaceSource) is sampled once for the D step and once for the G step like the reference
        (train.py:246-251, 303-306); a fixed `mesh` tuple is used for both otherwise.  With log=False the
        packed scalar all-reduce is still issued but nothing is copied to the host (returns tensors)."""
        a = self.args
        dev = self.device
        batch = real_img.shape[0]
        i = self.iteration
        losses = {}
        g, d = self.generator, self.discriminator
        # ---- D  (G only produces samples here: plain module under no_grad, no DDP bookkeeping —
        # a DDP forward that is never followed by a backward breaks the reducer's iteration state)
        requires_grad(d, True)
        noise = mixing_noise(batch, a["latent"], a["mixing"], dev, self.np_rng)
        with torch.no_grad():
            fake_img, _, _ = self._generate(g, noise, self._mesh(mesh, faces, batch))
            fake_img = self._augment(fake_img)
            real_aug = self._augment(real_img)
        # one discriminator pass over both halves.  Interleaving keeps the minibatch-stddev groups of the two
        # separate calls of the reference (model.py:325-332 views the batch as [group, batch // group]: sample b
        # falls into sub-batch b % 2), so fake and real statistics never mix and the outputs are identical
        # — which needs whole groups: batch % stddev_group == 0.  Smaller / ragged batches (the reference then uses
        # group = min(batch, 4)) take the reference's two calls.
        fake_pred, real_pred = d_fake_real(self.d_ddp, d, fake_img, real_aug)
        d_loss = d_logistic_loss(real_pred, fake_pred)
        losses["d"] = d_loss
        losses["real_score"] = real_pred.mean()
        losses["fake_score"] = fake_pred.mean()
        d.zero_grad(set_to_none=True)
        d_loss.backward()
        self.d_optim.step()
        if a["augment"] and a["augment_p"] <= 0:
            # ADA: p follows the sign statistics of D(real), all-reduced over ranks (train.py:269-280)
            stat = torch.stack([torch.sign(real_pred.detach()).sum(),
                                torch.tensor(float(real_pred.shape[0]), device=dev)])
            self.ada_augment += sr_dist.reduce_sum(stat)
            if float(self.ada_augment[1]) > 255:
                pred_signs, n_pred = self.ada_augment.tolist()
                self.r_t_stat = pred_signs / n_pred
                sign = 1 if self.r_t_stat > a["ada_target"] else -1
                self.ada_aug_p = min(1.0, max(0.0, self.ada_aug_p + sign * a["ada_target"] / a["ada_length"] * n_pred))
                self.ada_augment.mul_(0)
        if i % a["d_reg_every"] == 0:
            real_req = real_img.detach().requires_grad_(True)
            real_pred = self.d_ddp(real_req)
            r1_loss = d_r1_loss(real_pred, real_req)
            d.zero_grad(set_to_none=True)
            (a["r1"] / 2 * r1_loss * a["d_reg_every"] + 0 * real_pred[0]).backward()
            self.d_optim.step()
            losses["r1"] = r1_loss
        # ---- G  (D is a fixed critic here: its plain module with frozen parameters)
        requires_grad(d, False)
        noise = mixing_noise(batch, a["latent"], a["mixing"], dev, self.np_rng)
        g_mesh = self._mesh(mesh, faces, batch)
        fake_img, _, _ = self._generate(self.g_ddp, noise, g_mesh)
        g_loss = g_nonsaturating_loss(d(self._augment(fake_img)))
        losses["g"] = g_loss
        g.zero_grad(set_to_none=True)
        g_loss.backward()
        self.g_optim.step()
        if i % a["g_reg_every"] == 0:
            pb = max(1, batch // a["path_batch_shrink"]) if a["path_batch_shrink"] else batch
            noise = mixing_noise(pb, a["latent"], a["mixing"], dev, self.np_rng)
            sub_mesh = None
            if g_mesh is not None:
                sub_mesh = (g_mesh[0][:pb].detach().requires_grad_(True),
                            g_mesh[1][:pb].detach().requires_grad_(True), g_mesh[2])
            fake_img, latents, normals = self._generate(self.g_ddp, noise, sub_mesh, return_latents=True,
                                                        return_normals=True)
            targets = [latents] + (list(normals) if normals else [])
            # the EMA of the path length stays rank-local like the reference's (train.py:345-346); only
            # its logged value is averaged over ranks (train.py:353-354)
            path_loss, self.mean_path_length, path_lengths = g_path_regularize(
                fake_img, targets, self.mean_path_length)
            g.zero_grad(set_to_none=True)
            weighted = a["path_regularize"] * a["g_reg_every"] * path_loss
            if a["path_batch_shrink"]:
                weighted = weighted + 0 * fake_img[0, 0, 0, 0]
            weighted.backward()
            self.g_optim.step()
            losses["path"] = path_loss
            losses["path_length"] = path_lengths.mean()
            losses["mean_path"] = self.mean_path_length
        accumulate(self.g_ema, g, self.accum)
        self.iteration += 1
        return sr_dist.reduce_scalars(losses, to_host=log)   # one collective (+ one host read when logging)

    # ---- checkpoint contract of the reference (train.py:411-420): see checkpoint.py
    def state_dict(self):
        from . import checkpoint

        return checkpoint.trainer_state(self)


class SyntheticImages:
    """In-memory stand-in for the reference's LMDB dataset (reference dataset.py:56-92): float
    images in [-1, 1], random horizontal flip, sharded by rank.  Sampling and flipping run on the device
    (a seeded generator of that device), so fetching a batch never synchronises the host."""

    def __init__(self, n, size, device, seed=1234):
        g = torch.Generator(device="cpu").manual_seed(seed + sr_dist.get_rank())
        self.data = (torch.rand(n, 3, size, size, generator=g) * 2 - 1).to(device)
        self.gen = torch.Generator(device=self.data.device).manual_seed(seed + 1 + sr_dist.get_rank())

    def batch(self, b):
        dev = self.data.device
        idx = torch.randint(0, self.data.shape[0], (b,), device=dev, generator=self.gen)
        x = self.data[idx]
        flip = torch.rand(b, device=dev, generator=self.gen) < 0.5
        return torch.where(flip[:, None, None, None], x.flip(3), x)


def synthetic_mesh(batch, device, seed=0, face_sized=True):
    v0, tri = synth.face_sized_mesh() if face_sized else synth.uv_ellipsoid(16, 14)
    v = synth.random_poses(v0, batch, seed=seed)
    nrm = synth.vertex_normals(v, tri)
    return (torch.from_numpy(v).to(device), torch.from_numpy(nrm).to(device), torch.from_numpy(tri).to(device))


def smooth_basis(v0, dim, key, amplitude):
    """[dim, nv*3] deformation basis whose rows are LOW-FREQUENCY displacement fields of the mean mesh
    (direction_k * sin(2 pi f_k . v + phase_k), |f_k| <= 1.5 cycles per unit): neighbouring vertices move
    together, as the PCA modes of a real 3DMM do.  (Independent per-vertex noise — e.g. the reference's own
    random initialisation, face_model.py:16-19 — turns the surface into a triangle soup with ~20 px triangles
    and 100x overdraw, which no face mesh produces.)"""
    freq = synth.det_uniform((dim, 3), key) * 1.5
    phase = synth.det_uniform((dim,), key + 1) * np.pi
    direc = synth.det_normal((dim, 3), key + 2)
    direc /= np.maximum(np.linalg.norm(direc, axis=1, keepdims=True), 1e-6)
    field = np.sin(2 * np.pi * (v0.astype(np.float64) @ freq.T.astype(np.float64)) + phase[None])      # [nv, dim]
    basis = field.T[:, :, None] * direc[:, None, :].astype(np.float64)                                    # [dim, nv, 3]
    return (basis.reshape(dim, -1) * amplitude).astype(np.float32)


class SyntheticFaceSource:
    """Per-iteration mesh sampling of the reference's training loop (train.py:246-251: coefficients ->
    3DMM vertices -> random pose -> vertex normals), with a synthetic 3DMM: the face-sized UV mesh as the
    mean (BFM itself is licensed and absent) and smooth shape / expression bases (`smooth_basis`) scaled so
    that a sample deforms the surface by a few percent of its size.  Everything runs on the device: one
    GEMM, one small matmul, one gather kernel (utils_3d.mesh_point_normal)."""

    def __init__(self, device, shape_dim=80, expression_dim=64, seed=0, face_sized=True):
        from . import face_model, utils_3d

        v0, tri = synth.face_sized_mesh() if face_sized else synth.uv_ellipsoid(16, 14)
        nv = v0.shape[0]
        state = np.random.get_state()
        np.random.seed(seed)
        try:
            # coefficient sigmas are the reference's defaults (shape 1, expression 0.01, face_model.py:9-10)
            wsh = smooth_basis(v0, shape_dim, 901, 0.05 / np.sqrt(max(shape_dim, 1)))
            wex = smooth_basis(v0, expression_dim, 905, 3.0 / np.sqrt(max(expression_dim, 1)))
            self.model = face_model.LinearMorphableModel(nv, shape_dim, expression_dim, v0, wsh, wex).to(device)
        finally:
            np.random.set_state(state)
        self.tri = torch.from_numpy(tri.astype(np.int64)).to(device)
        self.device = device
        self._u = utils_3d
        # pose sigmas (the defaults of reference utils_3d.py:360) resident on the device: sampling a batch then issues
        # no host-to-device copy at all (coefficients, poses and normals are drawn / computed there)
        self.pose_sigma = torch.tensor([.5, .1, .05, .1, .1, .1, .15], dtype=torch.float32, device=device)

    @torch.no_grad()
    def sample(self, batch):
        coeff = self.model.random_input(batch)              # torch.normal on the model's device
        vert = self._u.random_apply_pose3D(p=self.pose_sigma, v=self.model(coeff))
        return vert, self._u.mesh_point_normal(vert, self.tri), self.tri


def main():
    ap = argparse.ArgumentParser(description="StyleRenderer training step on synthetic data")
    ap.add_argument("--iter", type=int, default=16)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--latent", type=int, default=512)
    ap.add_argument("--n_mlp", type=int, default=8)
    ap.add_argument("--r1", type=float, default=10)
    ap.add_argument("--path_regularize", type=float, default=2)
    ap.add_argument("--path_batch_shrink", type=int, default=2)
    ap.add_argument("--d_reg_every", type=int, default=16)
    ap.add_argument("--g_reg_every", type=int, default=4)
    ap.add_argument("--mixing", type=float, default=0.9)
    ap.add_argument("--lr", type=float, default=0.002)
    ap.add_argument("--channel_multiplier", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mesh", action="store_true", help="GeneratorWithMap + rasterised normal maps")
    ap.add_argument("--augment", action="store_true", help="adaptive discriminator augmentation")
    ap.add_argument("--augment_p", type=float, default=0)
    ap.add_argument("--ada_target", type=float, default=0.6)
    ap.add_argument("--ada_length", type=int, default=500 * 1000)
    ap.add_argument("--ckpt", type=str, default=None, help="checkpoint to resume from (reference layout)")
    ap.add_argument("--save", type=str, default=None, help="write a checkpoint here after the last iteration")
    ap.add_argument("--graphs", action="store_true",
                    help="replay every phase from a hipGraph (graph_train.GraphedTrainer: GPU only, no --augment)")
    ap.add_argument("--log_every", type=int, default=1, help="print (and read back) the losses every N iterations")
    args = ap.parse_args()
    rank, _, world, device = sr_dist.initialize(seed=args.seed)
    targs = (args.size, args.latent, args.n_mlp, args.channel_multiplier, args.lr, args.r1, args.path_regularize,
             args.path_batch_shrink, args.d_reg_every, args.g_reg_every, args.mixing, args.mesh, device, args.seed,
             args.augment, args.augment_p, args.ada_target, args.ada_length)
    faces = SyntheticFaceSource(device, seed=args.seed) if args.mesh else None
    if args.graphs:
        from . import graph_train

        tr = graph_train.GraphedTrainer(*targs, batch=args.batch,
                                        mesh_vertices=faces.model.dim[2] // 3 if faces is not None else None)
    else:
        tr = Trainer(*targs)
    if args.ckpt:
        from . import checkpoint

        checkpoint.load_checkpoint(args.ckpt, tr, map_location=device)
    data = SyntheticImages(max(64, args.batch * 4), args.size, device)
    t0 = None
    for it in range(args.iter):
        if it == 1:
            if device.type == "cuda":
                torch.cuda.synchronize()
            t0 = time.perf_counter()
        log = it % max(1, args.log_every) == 0 or it == args.iter - 1
        out = tr.step(data.batch(args.batch), faces=faces, log=log)
        if rank == 0 and log:
            print("iter %d  " % it + "  ".join("%s %.4f" % (k, float(v)) for k, v in sorted(out.items())), flush=True)
    if device.type == "cuda":
        torch.cuda.synchronize()
    if rank == 0 and t0 is not None and args.iter > 1:
        dt = time.perf_counter() - t0
        print("%.2f img/s over %d GPUs (%d timed iterations)" % (args.batch * world * (args.iter - 1) / dt,
                                                              world, args.iter - 1))
    if args.save and rank == 0:
        from . import checkpoint

        checkpoint.save_checkpoint(args.save, tr)


if __name__ == "__main__":
    main()
