"""Latent inversion of a face image (BASELINE config[4]).

The reference has no fitting script; its pieces are the generator's `input_is_latent` entry
(reference model.py:224-295), the differentiable rasterizer (op/rasterize.py:17-82) and the LPIPS metric
(lpips/networks_basic.py:27-92).  This module defines the loop BASELINE.json names:

    variables   w [1, n_latent, style_dim]  (W+ latent, initialised at the mean latent)
                pose [7] = yaw, pitch, roll (rad), tx, ty, tz, log-scale of the 3DMM mesh
    forward     vertices = v0 @ (exp(s) * R(yaw, pitch, roll)) + t ; normals = n0 @ R
                image = GeneratorWithMap([w], (vertices, normals, tri), input_is_latent=True, noise=fixed)
    loss        LPIPS-shaped distance(image, target) + pixel_weight * mean((image - target)^2)
    update      Adam, `steps` iterations (default 400)

The pose gradient reaches the vertices only through the rasterizer's backward (deterministic gather, so two
runs from the same state produce bit-identical trajectories).  On a GPU the whole iteration — forward, backward,
Adam — is one hipGraph (batch 1 is launch-latency bound: ~2 000 kernels per step), replayed `steps` times; the
loss history stays on the device until the end.
"""
import torch
from torch import optim

from . import graphs, utils_3d


class _FlatAdamSet:
    """The per-variable optim.FlatAdam instances of the device path behind torch.optim's two calls (scripts and callers
    written against `inverter.optim.step()` keep working: ADVICE r4)."""

    def __init__(self, adams):
        self.adams = adams

    @torch.no_grad()
    def step(self):
        for p, adam in self.adams:
            if p.grad is not None:
                adam.flat_g[:p.numel()].copy_(p.grad.reshape(-1))
                adam.step()

    def zero_grad(self, set_to_none=True):
        for p, _ in self.adams:
            p.grad = None


class LatentInverter:
    def __init__(self, generator, perceptual, target, mesh, lr=0.05, pose_lr=0.01, pixel_weight=1.0, noise=None,
                 n_mean_latent=4096, use_graph=None, optimise_pose=True):
        self.g = generator.eval()
        self.perceptual = perceptual.eval()
        for p in list(self.g.parameters()) + list(self.perceptual.parameters()):
            p.requires_grad_(False)
        # frozen networks: tap-major weights, demodulation matrices and adjoints are prepared once, not per step
        from .op.weight_prep import freeze_prepared_weights

        freeze_prepared_weights(self.g)
        self.device = target.device
        self.target = target.detach()
        self.v0, self.n0, self.tri = (t.detach() for t in mesh)
        self.pixel_weight = float(pixel_weight)
        self.with_map = hasattr(self.g, "norm_to_style")
        with torch.no_grad():
            mean_w = self.g.mean_latent(n_mean_latent)                                  # [1, D]
            self.target_feats = [f.detach() for f in self.perceptual.features(self.target)]
        self.w = mean_w.unsqueeze(1).repeat(1, self.g.n_latent, 1).clone().requires_grad_(True)
        self.pose = torch.zeros(7, device=self.device, requires_grad=optimise_pose)
        self.noise = noise if noise is not None else [n.detach() for n in self.g.make_noise()]
        on_gpu = self.device.type == "cuda"
        groups = [{"params": [self.w], "lr": lr}]
        if optimise_pose:
            groups.append({"params": [self.pose], "lr": pose_lr})
        if on_gpu:
            # one sr_adam_flat launch per variable (optim.FlatAdam, the training loop's optimiser) instead of the ~30
            # multi-tensor passes of torch's capturable foreach Adam: at batch 1 every launch is ~5 us of an 8 ms step
            from .optim import ALIGN, FlatAdam

            self._adams = []
            for g_ in groups:
                p = g_["params"][0]
                flat_g = torch.zeros((p.numel() + ALIGN - 1) // ALIGN * ALIGN, device=self.device)
                self._adams.append((p, FlatAdam([p], flat_g, lr=g_["lr"], betas=(0.9, 0.999))))
            self.optim = _FlatAdamSet(self._adams)       # `.optim.step()` / `.zero_grad()` like torch's optimiser
        else:
            self._adams = None
            self.optim = optim.Adam(groups, betas=(0.9, 0.999))
        self.use_graph = on_gpu if use_graph is None else bool(use_graph)
        self.graph = None
        self.loss_value = torch.zeros((), device=self.device)
        self.image = None

    # ---- model ----------------------------------------------------------------------------------------
    def posed_mesh(self):
        lin, rot = utils_3d.pose_matrices(self.pose)                                    # [1, 3, 3] each
        # [nv, 3] x [3, 3]: one streaming kernel each (utils_3d.affine3), not a 3-wide library GEMM
        v = utils_3d.affine3(self.v0, lin, self.pose[3:6].view(1, 3))
        n = utils_3d.affine3(self.n0, rot)
        return v.contiguous(), n.contiguous(), self.tri

    def render(self):
        if self.with_map:
            img, _, _ = self.g([self.w], self.posed_mesh(), input_is_latent=True, noise=self.noise)
        else:
            img, _ = self.g([self.w], input_is_latent=True, noise=self.noise)
        return img

    def loss(self, img):
        from .op.lpips_layer import mse

        d = self.perceptual.distance_to(self.target_feats, img).mean()
        return d + self.pixel_weight * mse(img, self.target)

    def _iteration(self):
        self.w.grad = None
        self.pose.grad = None
        img = self.render()
        value = self.loss(img)
        value.backward()
        self.optim.step()
        self.loss_value.copy_(value.detach())
        self.image = img.detach()

    # ---- driver ---------------------------------------------------------------------------------------
    def _warm_and_capture(self, history, warmup=3):
        """`warmup` real iterations on a side stream (lazy initialisation, Adam state), then one capture.
        Returns the number of optimisation steps already taken (capture itself executes nothing)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for k in range(warmup):
                self._iteration()
                history[k] = self.loss_value
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = graphs.capture(self._iteration)      # memset nodes of torch's reductions repaired: graphs.py
        return warmup

    def run(self, steps=400):
        """Runs `steps` Adam iterations; returns the loss history [steps] (device tensor: no host read here)."""
        history = torch.zeros(steps, device=self.device)
        done = 0
        if self.use_graph and self.graph is None and steps > 4:
            done = self._warm_and_capture(history)
        for i in range(done, steps):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._iteration()
            history[i] = self.loss_value
        return history
