"""Adam over flat buffers (the optimiser of reference train.py:529-536, `optim.Adam(..., lr * ratio, betas = (0 ** ratio,
0.99 ** ratio))`) for the graph-replayed training iteration.

`FlatAdam` moves its parameters into ONE contiguous fp32 buffer (each parameter becomes a 256-byte aligned view of it,
so every kernel that consumes a weight still sees an aligned tensor) and keeps the gradients, first and second
moments in buffers of the same layout; a step is one launch of `sr_adam_flat` (28 B per parameter) instead of the
~10 multi-tensor passes of torch's foreach Adam.  `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format
(per-parameter `step`, `exp_avg`, `exp_avg_sq`), so checkpoints move between the two.
"""
import torch

from . import _lib

ALIGN = 64          # floats: 256-byte aligned views


def flat_layout(params, multiple_of=1, order=None):
    """Offsets of 256-byte aligned slots and the total length, rounded up to `multiple_of` * ALIGN floats (so the
    buffer splits evenly over the ranks of a reduce-scatter).  `order` (a permutation of range(len(params))) places
    the slots in that order — graph_train lays the buffer out in gradient ARRIVAL order so that contiguous buckets
    complete one after the other during the backward; `offs[i]` is always the offset of `params[i]`."""
    offs, off = [0] * len(params), 0
    for i in (order if order is not None else range(len(params))):
        offs[i] = off
        off += (params[i].numel() + ALIGN - 1) // ALIGN * ALIGN
    step = ALIGN * max(1, int(multiple_of))
    return offs, (off + step - 1) // step * step


def flat_views(flat, params, offs):
    return [flat[o:o + p.numel()].view_as(p) for p, o in zip(params, offs)]


class FlatAdam:
    def __init__(self, params, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, offs=None):
        self.params = list(params)
        p0 = self.params[0]
        if any(p.dtype != torch.float32 for p in self.params):
            raise RuntimeError("FlatAdam: float32 parameters required")
        self.offs = list(offs) if offs is not None else flat_layout(self.params)[0]
        self.total = flat_grad.numel()                 # may carry tail padding (rank-divisible length)
        if self.total < max(o + p.numel() for o, p in zip(self.offs, self.params)):
            raise RuntimeError("FlatAdam: gradient buffer does not follow flat_layout(params)")
        self.flat_g = flat_grad
        self.flat_p = torch.zeros(self.total, device=p0.device)
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.step_t = torch.zeros((), device=p0.device)
        with torch.no_grad():
            for p, view in zip(self.params, flat_views(self.flat_p, self.params, self.offs)):
                view.copy_(p)
                p.data = view                      # the module now reads its weights out of the flat buffer
        self.param_groups = [{"params": self.params, "lr": float(lr), "betas": (float(betas[0]), float(betas[1])),
                              "eps": float(eps), "weight_decay": 0, "amsgrad": False}]
        self.guards = None               # set_guards(): positions of flat_grad that must not be NaN for a step to apply
        self.skipped = None

    def set_guards(self, offsets):
        """Data-parallel training: `offsets` = the first element of every all-reduce bucket of the gradient buffer.
        A step whose gradient carries NaN at one of them is REFUSED on the device (p / m / v untouched) and
        `self.skipped` (pinned host word) is raised — see sr_adam_flat_guarded and distributed.BucketedGradReducer:
        a bucket signal lost on one rank becomes a refused step on every rank instead of an update with a half-written
        gradient.  A refused step does not advance the step count either.  NaN semantics: ONLY these positions are
        sampled — NaN elsewhere in the buffer is not detected, and a gradient that is legitimately NaN at a guard position
        refuses the step too (training on NaN gradients is lost anyway)."""
        import ctypes

        offsets = [int(o) for o in offsets]
        if len(offsets) > 16:
            raise ValueError("FlatAdam.set_guards: at most 16 guard positions")
        self.guards = (ctypes.c_int64 * len(offsets))(*offsets)
        if self.flat_p.device.type == "cuda":
            self.skipped = torch.zeros(1, dtype=torch.int32).pin_memory()
        else:
            self.skipped = torch.zeros(1, dtype=torch.int32)
        return self.skipped

    def step(self):
        g = self.param_groups[0]
        if self.flat_p.device.type != "cuda":
            if self.guards is not None and bool(torch.isnan(self.flat_g[list(self.guards)]).any()):
                self.skipped.fill_(1)          # refused: p / m / v AND the step count stay as they were
                return
            self.step_t.add_(1.0)
            return self._step_host(g)
        # (device: the guarded kernel takes this increment back when it refuses the step)
        self.step_t.add_(1.0)
        if self.guards is not None:
            rc = _lib.lib().sr_adam_flat_guarded(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(),
                                                 self.v.data_ptr(), self.total, g["lr"], g["betas"][0], g["betas"][1],
                                                 g["eps"], self.step_t.data_ptr(), self.guards, len(self.guards),
                                                 self.skipped.data_ptr(), _lib.current_stream(self.flat_p.device))
            _lib.check(rc, "sr_adam_flat_guarded")
            return
        rc = _lib.lib().sr_adam_flat(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(),
                                     self.v.data_ptr(), self.total, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                     self.step_t.data_ptr(), _lib.current_stream(self.flat_p.device))
        _lib.check(rc, "sr_adam_flat")

    @torch.no_grad()
    def _step_host(self, g):
        """CPU tensors (gloo tests, plumbing): the arithmetic of k_adam_flat (csrc/fused_elem.hip) in torch ops."""
        b1, b2 = g["betas"]
        t = float(self.step_t)
        bc1 = 1.0 - b1 ** t
        bc2s = (1.0 - b2 ** t) ** 0.5
        grad = self.flat_g
        self.m.add_((grad - self.m) * (1.0 - b1))
        self.v.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
        self.flat_p.sub_((g["lr"] / bc1) * (self.m / (self.v.sqrt() / bc2s + g["eps"])))

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()

    # ---- torch.optim.Adam's checkpoint format ---------------------------------------------------------
    def state_dict(self):
        state = {}
        if float(self.step_t) > 0:
            for i, (m, v) in enumerate(zip(flat_views(self.m, self.params, self.offs),
                                           flat_views(self.v, self.params, self.offs))):
                state[i] = {"step": self.step_t.detach().clone(), "exp_avg": m.detach().clone(),
                            "exp_avg_sq": v.detach().clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError("FlatAdam.load_state_dict: parameter count mismatch")
        for k in ("lr", "betas", "eps"):
            if k in group:
                self.param_groups[0][k] = tuple(float(b) for b in group[k]) if k == "betas" else float(group[k])
        step = 0.0
        with torch.no_grad():
            self.m.zero_()
            self.v.zero_()
            mv = flat_views(self.m, self.params, self.offs)
            vv = flat_views(self.v, self.params, self.offs)
            for i, st in sd["state"].items():
                mv[int(i)].copy_(st["exp_avg"])
                vv[int(i)].copy_(st["exp_avg_sq"])
                step = max(step, float(st["step"]))
            self.step_t.fill_(step)
