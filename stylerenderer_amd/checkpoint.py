"""Checkpoint contract of the reference.

  save   reference train.py:411-420   torch.save({'g','d','g_ema','g_optim','d_optim','args','ada_aug_p'})
         to checkpoint/%06d.pt every 10 000 iterations
  resume reference train.py:537-556   start iteration parsed from the file name, every key optional
  sample reference generate.py:61-69  Generator <- ckpt['g_ema'] (else ckpt['g'])

The state_dict keys of Generator / GeneratorWithMap / Discriminator equal the reference's (165 keys at
256x256 incl. the duplicated `to_rgbs` tail), so checkpoints travel in both directions.
One difference is stated rather than hidden: this repo's G optimiser skips the never-used ToRGB tail, so
a reference `g_optim` (which also holds Adam moments for those dead parameters) is re-indexed by
parameter name on load, and written back in the reference's full indexing on save.
"""
import os
import re

import torch

CKPT_KEYS = ("g", "d", "g_ema", "g_optim", "d_optim", "args", "ada_aug_p")


def checkpoint_name(iteration, total_iter=800000):
    """'%06d.pt'-style name: digits = max(floor(log10(iter)) + 1, 6) (reference train.py:236-237)."""
    import math

    bits = max(int(math.floor(math.log(max(total_iter, 1)) / math.log(10))) + 1, 6)
    return ("%%0%dd.pt" % bits) % iteration


def start_iter_from_name(path):
    """reference train.py:540-544: int(basename without extension), 0 when it is not a number."""
    stem = os.path.splitext(os.path.basename(path))[0]
    return int(stem) if re.fullmatch(r"\d+", stem) else 0


def _full_optim_state(optimizer, module, used_names):
    """Adam state re-indexed over ALL parameters of `module` in registration order (the reference's
    optimiser is built from generator.parameters()); parameters this repo does not optimise get no entry."""
    sd = optimizer.state_dict()
    names = [n for n, _ in module.named_parameters()]
    pos = {n: i for i, n in enumerate(names)}
    state = {pos[n]: sd["state"][i] for i, n in enumerate(used_names) if i in sd["state"]}
    group = dict(sd["param_groups"][0])
    group["params"] = list(range(len(names)))
    return {"state": state, "param_groups": [group]}


def _subset_optim_state(full, module, used_names, optimizer):
    names = [n for n, _ in module.named_parameters()]
    pos = {n: i for i, n in enumerate(names)}
    state = {}
    for i, n in enumerate(used_names):
        if pos[n] in full["state"]:
            state[i] = full["state"][pos[n]]
    group = dict(full["param_groups"][0])
    group["params"] = list(range(len(used_names)))
    # hyper-parameters of the running optimiser win over stale keys a different torch version wrote
    cur = optimizer.state_dict()["param_groups"][0]
    for k, v in cur.items():
        group.setdefault(k, v)
    return {"state": state, "param_groups": [group]}


def trainer_state(trainer):
    """The reference's checkpoint dict for a train.Trainer."""
    g = trainer.generator
    used = [n for n, _ in g.named_parameters() if n not in trainer.frozen]
    def own(module):          # parameters may be views of a flat optimiser buffer: save compact copies
        return {k: v.detach().clone() for k, v in module.state_dict().items()}

    out = {
        "g": own(g), "d": own(trainer.discriminator), "g_ema": own(trainer.g_ema),
        "g_optim": _full_optim_state(trainer.g_optim, g, used),
        "d_optim": trainer.d_optim.state_dict(),
        "args": dict(trainer.args), "ada_aug_p": float(trainer.ada_aug_p),
        # extensions the reference's loaders ignore: exact resume without parsing the file name
        "iteration": int(trainer.iteration), "mean_path_length": float(trainer.mean_path_length),
    }
    return out


def save_checkpoint(path, trainer):
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    tmp = path + ".tmp"
    torch.save(trainer_state(trainer), tmp)
    os.replace(tmp, path)                       # a killed job never leaves a truncated checkpoint behind
    return path


def load_checkpoint(path, trainer, map_location="cpu", strict=True):
    """Resume semantics of reference train.py:537-556: every key is optional; `g_ema` falls back to `g`;
    the start iteration comes from the file name unless the checkpoint carries one.

    COLLECTIVE for a graph_train.GraphedTrainer at world size > 1: EVERY rank must call it with the same file (the reference does: train.py:537 runs on
    all ranks).  A GraphedTrainer drops its captured graphs here and re-captures on the next step(); the warm-up
    iterations of that re-capture issue gradient collectives, so a "rank 0 loads and broadcasts" flow would leave the
    other ranks without matching calls.  The check below turns that mistake into an error (or a hang AT THIS LINE
    that the RCCL watchdog names) instead of a hang somewhere inside the next step."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    _assert_collective_load(trainer, int(ckpt.get("iteration", start_iter_from_name(path))))
    g = trainer.generator
    if "g" in ckpt:
        g.load_state_dict(ckpt["g"], strict=strict)
    if "d" in ckpt:
        trainer.discriminator.load_state_dict(ckpt["d"], strict=strict)
    if "g_ema" in ckpt:
        trainer.g_ema.load_state_dict(ckpt["g_ema"], strict=strict)
    elif "g" in ckpt:
        trainer.g_ema.load_state_dict(ckpt["g"], strict=strict)
    used = [n for n, _ in g.named_parameters() if n not in trainer.frozen]
    if "g_optim" in ckpt:
        full = ckpt["g_optim"]
        n_all = sum(1 for _ in g.parameters())
        if len(full["param_groups"][0]["params"]) == n_all:
            trainer.g_optim.load_state_dict(_subset_optim_state(full, g, used, trainer.g_optim))
        else:
            trainer.g_optim.load_state_dict(full)
    if "d_optim" in ckpt:
        trainer.d_optim.load_state_dict(ckpt["d_optim"])
    trainer.ada_aug_p = float(ckpt.get("ada_aug_p", trainer.ada_aug_p))
    trainer.iteration = int(ckpt.get("iteration", start_iter_from_name(path)))
    if "mean_path_length" in ckpt:
        # in place: graph_train's path-length graph was captured on this very tensor
        with torch.no_grad():
            trainer.mean_path_length.fill_(float(ckpt["mean_path_length"]))
    if getattr(trainer, "graphs", None):
        # captured graphs bake the optimiser's lr / betas and the Adam step counter's address: re-capture on the next
        # step() (build_graphs restores the loaded state after its warm-up iterations)
        trainer.graphs = {}
    return ckpt


def _assert_collective_load(trainer, iteration):
    from . import distributed as sr_dist

    # only a trainer whose next step issues collectives of its own accord (graph_train.GraphedTrainer at world > 1:
    # the re-capture warm-up) — loading into a plain, non-distributed object on one rank is legitimate
    if sr_dist.get_world_size() <= 1 or getattr(trainer, "world", 1) <= 1 or not hasattr(trainer, "graphs"):
        return
    dev = getattr(trainer, "device", torch.device("cpu"))
    mine = torch.tensor([float(iteration), -float(iteration)], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(mine, op=torch.distributed.ReduceOp.MAX)
    lo, hi = -float(mine[1]), float(mine[0])
    if lo != hi:
        raise RuntimeError("load_checkpoint must load the SAME checkpoint on every rank: start iterations %d..%d "
                           "differ (rank %d has %d)" % (lo, hi, sr_dist.get_rank(), iteration))


def load_generator(path, size, latent=512, n_mlp=8, channel_multiplier=2, device="cpu", with_map=False):
    """generate.py:56-69: build the generator, load 'g_ema' (else 'g'), eval mode."""
    from .model import Generator, GeneratorWithMap

    cls = GeneratorWithMap if with_map else Generator
    g = cls(size, latent, n_mlp, channel_multiplier=channel_multiplier)
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "g_ema" in ckpt:
        g.load_state_dict(ckpt["g_ema"])
    elif "g" in ckpt:
        g.load_state_dict(ckpt["g"])
    else:
        raise KeyError("checkpoint %s holds neither 'g_ema' nor 'g'" % path)
    return g.to(device).eval()
