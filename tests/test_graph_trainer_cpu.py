"""CPU, world_size = 2 over gloo: graph_train.GraphedTrainer — the trainer bench.py's config[2] leg and the driver's
SCALE run use — executed with capture=False (same phases, hooks, buckets and collectives; hipGraph capture is the
only thing a GPU adds).  Replicas stay bit-identical through D / R1 / G / path-length phases, the reduced flat buffer
is the mean of the per-rank gradients, buckets are issued while the backward (incl. the path-length double backward)
is still running, and one-bucket "after the backward" mode agrees with the overlapped mode."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from stylerenderer_amd import checkpoint, graph_train, model, synth, train
from stylerenderer_amd import distributed as sr_dist

SIZE, LATENT, NMLP, BATCH = 8, 32, 2, 4


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def make_trainer(n_buckets=3, seed=3, **kw):
    v0, _ = synth.uv_ellipsoid(16, 14)
    return graph_train.GraphedTrainer(size=SIZE, latent=LATENT, n_mlp=NMLP, device="cpu", seed=seed, use_mesh=True,
                                      batch=BATCH, mesh_vertices=v0.shape[0], capture=False, n_buckets=n_buckets, **kw)


def phase_flat(tr, name, reducer, flat, seed):
    """Runs one phase body from a fixed RNG state; returns (flat gradient copy, reducer log, last-gradient stamp)."""
    torch.manual_seed(seed)
    tr.mean_path_length.fill_(0.25)               # the path phase updates this EMA: same start for every run
    stamp = {}
    first = tr.generator.style[1].weight if reducer is tr.reduce_g else tr.discriminator.convs[0][0].weight
    first.requires_grad_(True)                    # (the phase body sets the flags it needs anyway)
    h = first.register_hook(lambda g: stamp.__setitem__("last", time.perf_counter()))
    try:
        tr._bodies()[name]()
        reducer.wait()
    finally:
        h.remove()
    return flat.clone(), list(reducer.log), stamp.get("last")


def worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    sr_dist.initialize(backend="gloo")
    out = {}
    tr = make_trainer()
    out["buckets_g"] = [(b["lo"], b["hi"], len(b["members"])) for b in tr.reduce_g.buckets]
    out["describe"] = tr.reduce_g.describe()
    mesh = train.synthetic_mesh(BATCH, "cpu", seed=10 + rank, face_sized=False)
    data = train.SyntheticImages(16, SIZE, "cpu")
    # --- (1) two full iterations (i = 0 runs R1 and the path-length phase): replicas identical
    logs = [tr.step(data.batch(BATCH), mesh=mesh) for _ in range(2)]
    out["logs"] = logs
    out["checksum"] = torch.stack([p.detach().double().sum() for p in tr.generator.parameters()]
                                  + [p.detach().double().sum() for p in tr.discriminator.parameters()])
    out["flat_p"] = tr.g_optim.flat_p.clone()
    # --- (2) reduced flat buffer == mean of the per-rank gradients, for the G and the path-length phase; stamps
    for name, red, flat in (("g", tr.reduce_g, tr.flat_g), ("path", tr.reduce_g, tr.flat_g), ("d", tr.reduce_d, tr.flat_d),
                            ("r1", tr.reduce_d, tr.flat_d)):
        red.enabled = False
        local, _, _ = phase_flat(tr, name, red, flat, 100 + rank)
        red.enabled = True
        reduced, log, last = phase_flat(tr, name, red, flat, 100 + rank)
        both = [torch.zeros_like(local) for _ in range(world)]
        torch.distributed.all_gather(both, local)
        out["mean_" + name] = bool(torch.equal(reduced, (both[0] + both[1]) / world))
        out["nonzero_" + name] = float(reduced.abs().max())
        issues = [(b, t) for kind, b, t in log if kind == "issue"]
        out["stamps_" + name] = {"n": len(issues), "order": [b for b, _ in issues],
                                 "early": sum(t < last for _, t in issues) if last is not None else -1}
    # --- (3) one bucket, reduced after the backward (SR_GRAD_OVERLAP=0 mode) agrees with the overlapped mode
    tr1 = make_trainer(n_buckets=1)
    assert tr1.reduce_g.single is not None and tr1.reduce_g.single.mode == "allreduce"
    tr1._load_inputs(data.batch(BATCH), mesh, None)
    tr._load_inputs(tr1.s_real, mesh, None)
    for k in tr.s_inject:
        tr1.s_inject[k].copy_(tr.s_inject[k])     # same style-mixing crossover
    with torch.no_grad():
        tr1.g_optim.flat_p.zero_()
        for p1, p in zip(tr1.g_params, tr.g_params):
            p1.copy_(p)
        for p1, p in zip(tr1.d_params, tr.d_params):
            p1.copy_(p)
    a, _, _ = phase_flat(tr, "g", tr.reduce_g, tr.flat_g, 7 + rank)
    b, _, _ = phase_flat(tr1, "g", tr1.reduce_g, tr1.flat_g, 7 + rank)
    # same layout (arrival order is a property of the model), so the buffers compare element by element
    out["one_bucket_equal"] = bool(torch.equal(a, b))
    # --- (4) checkpoint out of the permuted flat layout, in the reference's indexing
    if rank == 0:
        path = checkpoint.save_checkpoint(os.path.join(outdir, checkpoint.checkpoint_name(2)), tr)
        eager = train.Trainer(size=SIZE, latent=LATENT, n_mlp=NMLP, device="cpu", seed=9, use_mesh=True, wrap_ddp=False)
        checkpoint.load_checkpoint(path, eager)
        names = [n for n, _ in tr.generator.named_parameters() if n not in tr.frozen]
        views_m = graph_train.flat_views(tr.g_optim.m, tr.g_params, tr.g_optim.offs)
        st = eager.g_optim.state_dict()["state"]
        out["ckpt_ok"] = all(torch.equal(st[i]["exp_avg"], views_m[i]) for i in range(len(names))) and all(
            torch.equal(p, q) for p, q in zip(eager.generator.parameters(), tr.generator.parameters()))
    torch.save(out, os.path.join(outdir, "rank%d.pt" % rank))
    sr_dist.synchronize()
    torch.distributed.destroy_process_group()


@pytest.fixture(scope="module")
def two_rank_run(tmp_path_factory):
    outdir = str(tmp_path_factory.mktemp("graphed"))
    mp.spawn(worker, args=(2, free_port(), outdir), nprocs=2, join=True)
    return [torch.load(os.path.join(outdir, "rank%d.pt" % r), weights_only=False) for r in range(2)]


def test_replicas_identical_after_all_four_phases(two_rank_run):
    r0, r1 = two_rank_run
    assert torch.equal(r0["checksum"], r1["checksum"]) and torch.equal(r0["flat_p"], r1["flat_p"])
    assert {"r1", "path", "path_length", "mean_path"} <= set(r0["logs"][0]) and "r1" not in r0["logs"][1]
    for log in r0["logs"]:
        assert all(np.isfinite(v) for v in log.values())
    assert r0["logs"] == r1["logs"]                                  # the packed scalar reduction


def test_flat_buffer_is_the_mean_of_the_rank_gradients(two_rank_run):
    for r in two_rank_run:
        for name in ("g", "path", "d", "r1"):
            assert r["mean_" + name], name
            assert r["nonzero_" + name] > 0, name


def test_buckets_are_issued_in_order_while_the_backward_runs(two_rank_run):
    r0, r1 = two_rank_run
    assert r0["buckets_g"] == r1["buckets_g"] and len(r0["buckets_g"]) == 3
    assert r0["describe"]["mode"] == "overlapped, 3 buckets"
    los = [b[0] for b in r0["buckets_g"]]
    assert los == sorted(los) and los[0] == 0
    for r in two_rank_run:
        for name in ("g", "path", "d", "r1"):
            st = r["stamps_" + name]
            assert st["n"] == 3 and st["order"] == [0, 1, 2], (name, st)
            # every bucket but the one that holds the last-produced gradients is on the wire before the backward ends
            assert st["early"] >= 2, (name, st)


def test_one_bucket_mode_agrees_and_checkpoint_uses_reference_indexing(two_rank_run):
    r0, r1 = two_rank_run
    assert r0["one_bucket_equal"] and r1["one_bucket_equal"]
    assert r0["ckpt_ok"]


# ---- single process ---------------------------------------------------------------------------------------------------
def test_small_batches_take_two_discriminator_calls():
    """the interleaved D pass equals the reference's two calls only for whole minibatch-stddev groups."""
    torch.manual_seed(0)
    d = model.Discriminator(SIZE)
    synth.fill_state_dict(d.state_dict(), salt=8)
    for b in (1, 2, 3, 4, 8):
        fake = torch.from_numpy(synth.det_normal((b, 3, SIZE, SIZE), 20 + b))
        real = torch.from_numpy(synth.det_normal((b, 3, SIZE, SIZE), 40 + b))
        calls = []

        def d_call(x):
            calls.append(x.shape[0])
            return d(x)

        fp, rp = train.d_fake_real(d_call, d, fake, real)
        assert calls == ([2 * b] if b % 4 == 0 else [b, b])
        want_f, want_r = d(fake), d(real)
        assert torch.allclose(fp, want_f, rtol=1e-5, atol=1e-6) and torch.allclose(rp, want_r, rtol=1e-5, atol=1e-6), b


def test_eager_graphed_trainer_single_process_trains_and_arrival_layout_is_a_permutation():
    tr = make_trainer()
    offs = tr.g_optim.offs
    assert sorted(offs) != offs                                       # arrival order differs from registration order
    spans = sorted((o, o + p.numel()) for o, p in zip(offs, tr.g_params))
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))        # slots do not overlap
    # the mapping network's gradients arrive last: they sit at the end of the buffer, in the last bucket
    names = [n for n, _ in tr.generator.named_parameters() if n not in tr.frozen]
    assert tr.reduce_g.bucket_of[names.index("style.1.weight")] == len(tr.reduce_g.buckets) - 1
    before = tr.g_optim.flat_p.clone()
    mesh = train.synthetic_mesh(BATCH, "cpu", seed=1, face_sized=False)
    log = tr.step(train.SyntheticImages(8, SIZE, "cpu").batch(BATCH), mesh=mesh)
    assert all(np.isfinite(v) for v in log.values()) and not torch.equal(before, tr.g_optim.flat_p)
    assert tr.reduce_g.describe()["mode"] == "off"                    # one rank: nothing to reduce


def test_guarded_flat_adam_refuses_a_marked_gradient_on_cpu():
    """optim.FlatAdam.set_guards + BucketedGradReducer.guard on CPU tensors (the host arithmetic of sr_adam_flat_guarded):
    NaN at a bucket's first element = the marker a timed-out bucket wait leaves (through the all-reduce, on every rank):
    the step is refused, parameters and moments stay, and the reducer's check() raises once."""
    import pytest
    import torch

    from stylerenderer_amd import distributed as sr_dist
    from stylerenderer_amd.optim import FlatAdam

    ps = [torch.nn.Parameter(torch.randn(64)) for _ in range(4)]
    flat = torch.ones(4 * 64)
    offs = [0, 64, 128, 192]
    views = [flat[i * 64:(i + 1) * 64] for i in range(4)]
    red = sr_dist.BucketedGradReducer(ps, views, offs, flat, world=1, n_buckets=2, force=True)
    opt = red.guard(FlatAdam(ps, flat, lr=0.1, offs=offs))
    assert list(opt.guards) == [b["lo"] for b in red.buckets]
    opt.step()
    red.check()
    p1, m1, t1 = opt.flat_p.clone(), opt.m.clone(), float(opt.step_t)
    flat[red.buckets[1]["lo"]] = float("nan")
    opt.step()
    assert torch.equal(opt.flat_p, p1) and torch.equal(opt.m, m1) and float(opt.step_t) == t1   # a refused step is no step
    with pytest.raises(RuntimeError, match="REFUSED"):
        red.check()
    red.check()
    flat.fill_(1.0)
    opt.step()
    assert not torch.equal(opt.flat_p, p1)
