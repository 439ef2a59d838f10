"""GPU: graph_train.GraphedTrainer at BASELINE config[2]'s size, and its overlapped gradient reduction on a ONE-rank
RCCL group (the only group a 1-GPU box can form): the signal nodes (sr_signal_set) inside the replayed backward, the
wait kernels (sr_signal_wait_timeout) on the communication stream and the RCCL calls between replays are exactly what
runs at N > 1."""
import os
import socket

import numpy as np
import pytest
import torch

from stylerenderer_amd import graph_train, train

pytestmark = pytest.mark.gpu
DEV = "cuda"


def full_size_trainer(**kw):
    dev = torch.device("cuda")
    faces = train.SyntheticFaceSource(dev, seed=0)
    tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, channel_multiplier=2, use_mesh=True, device=dev,
                                    seed=0, batch=4, mesh_vertices=faces.model.dim[2] // 3, **kw)
    return tr, faces, train.SyntheticImages(16, 256, dev)


def test_full_size_graphed_iteration_256_and_graph_equals_eager_for_every_phase():
    """BASELINE config[2] on the DEFAULT trainer of bench.py: GeneratorWithMap(256) + Discriminator(256), 4 images,
    iteration 0 (R1 + path-length double backward on batch 2) and two more; then, phase by phase, the replayed graph
    leaves the same flat gradient buffer as the same body launched eagerly from the same RNG state."""
    tr, faces, data = full_size_trainer()
    g0 = tr.g_optim.flat_p.clone()
    d0 = tr.d_optim.flat_p.clone()
    ema0 = [p.detach().clone() for p in tr.g_ema.parameters()]
    logs = [tr.step(data.batch(4), faces=faces) for _ in range(3)]
    assert set(tr.graphs) == {"d", "r1", "g", "path", "d_opt", "g_opt", "ema"}
    # the captured EMA (reference train.py:100-104, once per iteration) is what the eager one computes from here
    moved = [not torch.equal(a, p.detach()) for a, p in zip(ema0, tr.g_ema.parameters())]
    assert sum(moved) >= len(moved) - len(tr.frozen)
    before = [p.detach().clone() for p in tr.g_ema.parameters()]
    tr.graphs["ema"].replay()
    replayed = [p.detach().clone() for p in tr.g_ema.parameters()]
    with torch.no_grad():
        for p, b in zip(tr.g_ema.parameters(), before):
            p.copy_(b)
    train.accumulate(tr.g_ema, tr.generator, tr.accum)
    assert all(torch.equal(a, p.detach()) for a, p in zip(replayed, tr.g_ema.parameters()))
    # the meshes are drawn inside the D / G replays (a fresh batch each): the static buffers change without any eager copy
    m0 = tr.s_mesh["d"][0].clone()
    tr.graphs["d"].replay()
    torch.cuda.synchronize()
    assert not torch.equal(m0, tr.s_mesh["d"][0]) and torch.isfinite(tr.s_mesh["d"][1]).all()
    assert {"d", "g", "r1", "path", "path_length", "mean_path", "real_score", "fake_score"} <= set(logs[0])
    assert "r1" not in logs[1] and "path" not in logs[1]
    for log in logs:
        assert all(np.isfinite(v) for v in log.values()), log
    assert logs[0]["path_length"] > 0 and float(tr.mean_path_length) > 0
    assert float(tr.g_optim.step_t) == 4 and float(tr.d_optim.step_t) == 4      # warm-up steps are not counted
    assert tr.iteration == 3
    # every trainable parameter of both networks moved
    names = [n for n, _ in tr.generator.named_parameters() if n not in tr.frozen]
    for n, p, o in zip(names, tr.g_params, tr.g_optim.offs):
        assert not torch.equal(p.detach().reshape(-1), g0[o:o + p.numel()]), n
    for (n, _), p, o in zip(tr.discriminator.named_parameters(), tr.d_params, tr.d_optim.offs):
        assert not torch.equal(p.detach().reshape(-1), d0[o:o + p.numel()]), n
    dev = tr.device
    for name, flat in (("d", tr.flat_d), ("r1", tr.flat_d), ("g", tr.flat_g), ("path", tr.flat_g)):
        mpl = tr.mean_path_length.clone()
        state = torch.cuda.get_rng_state(dev)
        tr._bodies()[name]()
        eager = flat.clone()
        tr.mean_path_length.copy_(mpl)
        torch.cuda.set_rng_state(state, dev)
        flat.zero_()
        tr.graphs[name].replay()
        torch.cuda.synchronize()
        scale = float(eager.abs().max())
        err = float((flat - eager).abs().max())
        assert scale > 0 and err <= 1e-5 * scale, (name, err, scale)


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                            pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
    yield dist
    dist.destroy_process_group()


def test_overlapped_reduction_inside_the_replayed_backward(one_rank_group):
    """force_collectives=True on a one-rank RCCL group: every bucket's all-reduce is issued on the communication
    stream behind its event-record node.  (1) The trajectory equals the run without collectives bit for bit
    (all-reduce over one rank is the identity, / world = 1).  (2) Timing events: the first buckets of the
    path-length phase are reduced BEFORE the replay of that phase's backward has finished."""
    a, faces, data = full_size_trainer(force_collectives=True)
    b, _, _ = full_size_trainer()
    assert a.reduce_g.enabled and not b.reduce_g.enabled
    assert a.reduce_g.describe()["mode"] == "overlapped, 4 buckets"
    batches = [data.batch(4) for _ in range(3)]
    for tr in (a, b):
        torch.manual_seed(123)
        tr.np_rng = np.random.RandomState(5)
        meshes = [tuple(t.clone() for t in faces.sample(4)) for _ in range(3)]
        torch.manual_seed(321)
        tr.logs = [tr.step(x, mesh=m) for x, m in zip(batches, meshes)]
    assert torch.equal(a.g_optim.flat_p, b.g_optim.flat_p) and torch.equal(a.d_optim.flat_p, b.d_optim.flat_p)
    assert a.logs == b.logs
    for name in ("path", "d", "d", "path", "r1", "g"):
        ov = a.measure_overlap(name)
        print("overlap", ov)
        done = ov["bucket_done_ms_after_replay_end"]
        assert len(done) == 4 and done == sorted(done), ov
    for name in ("path", "d"):
        # bucket 0 (40 % of the bytes) is reduced while the backward is still running; the later buckets complete in
        # the fast low-resolution tail (G) / before the high-resolution layers (D).  (One-lane wait kernels compete
        # for a wave slot with 2 500 captured kernels: allow a retry before calling a late release a failure.)
        tries = [a.measure_overlap(name) for _ in range(3)]
        assert any(t["bucket_done_ms_after_replay_end"][0] < 0 for t in tries), (name, tries)
    # D phase: the heavy low-resolution weights are the FIRST gradients of the backward — bucket 0 must be reduced in
    # the first half of the replay, not merely before its end (a weight-gradient node that autograd schedules at the end
    # of the backward passes the weaker check above while destroying the overlap)
    assert any(t["bucket_done_ms_after_replay_end"][0] < -0.25 * t["replay_ms"] for t in tries), tries
    # bench.py's per-phase timing goes through the same arm + replay + issue + wait path on every rank: the signal
    # words hold the epoch of the last announced replay
    ms = a.time_phase("d", 2)
    assert ms > 0
    torch.cuda.synchronize()
    assert a.reduce_d.counters.cpu().tolist() == [a.reduce_d.epoch] * 4
    # a replay NOBODY announced (probes, an exception between replay and issue_all) republishes the epoch already
    # reached: the words do not run ahead of the host, so the next announced replay's collectives still wait for
    # their buckets — the LAST bucket cannot be reduced long before the replay that fills it ends (an incrementing
    # counter would release all four at the start of the replay: about -replay_ms)
    a.graphs["d"].replay()
    a.graphs["d"].replay()
    torch.cuda.synchronize()
    assert a.reduce_d.counters.cpu().tolist() == [a.reduce_d.epoch] * 4
    ov = a.measure_overlap("d")
    again = ov["bucket_done_ms_after_replay_end"]
    assert again == sorted(again)
    assert again[-1] > -0.1 * ov["replay_ms"], ov
    assert not a.reduce_d.status.any()
    a.reduce_d.check()


def test_lost_signal_times_out_and_raises(monkeypatch):
    """A wait whose signal never comes (here: an epoch nobody replays) gives up after SR_SIGNAL_TIMEOUT_S, stores the
    bucket id in pinned host memory, and the next host-side check raises with it instead of hanging."""
    from stylerenderer_amd import _lib
    from stylerenderer_amd import distributed as sr_dist

    monkeypatch.setenv("SR_SIGNAL_TIMEOUT_S", "0.2")
    ps = [torch.nn.Parameter(torch.zeros(64, device=DEV)) for _ in range(4)]
    flat = torch.zeros(4 * 64, device=DEV)
    views = [flat[i * 64:(i + 1) * 64] for i in range(4)]
    red = sr_dist.BucketedGradReducer(ps, views, [0, 64, 128, 192], flat, world=1, n_buckets=2, force=True)
    assert red.timeout_us == 200000
    red.arm()                              # epoch 1 announced, but no graph publishes it
    L = _lib.lib()
    _lib.check(L.sr_signal_wait_timeout(red.counters.data_ptr() + 4, 1, red.timeout_us, red.status.data_ptr() + 4, 2,
                                        red.comm.cuda_stream), "wait")
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match=r"bucket\(s\) \[1\]"):
        red.check()
    assert not red.status.any()            # reported once
    # a published epoch releases at once and leaves the status clean
    _lib.check(L.sr_signal_set(red.counters.data_ptr() + 4, red.epoch_dev.data_ptr(), _lib.current_stream()), "set")
    torch.cuda.synchronize()               # (the two kernels sit on different streams; a cold first launch of the set
    #                                         kernel — code object load — may take longer than the 0.2 s of this test)
    _lib.check(L.sr_signal_wait_timeout(red.counters.data_ptr() + 4, 1, red.timeout_us, red.status.data_ptr() + 4, 2,
                                        red.comm.cuda_stream), "wait")
    torch.cuda.synchronize()
    red.check()


def test_host_release_lost_signal_poisons_and_still_issues_the_collective(monkeypatch, one_rank_group):
    """in host-released mode a lost bucket signal must not raise BEFORE the collective is queued (the peers
    would sit in the all-reduce): the bucket's guard position gets NaN, the collective runs, the status word is set and
    check() raises afterwards — the same chain as the device mode's expired wait."""
    from stylerenderer_amd import distributed as sr_dist

    monkeypatch.setenv("SR_SIGNAL_TIMEOUT_S", "0.2")
    ps = [torch.nn.Parameter(torch.zeros(64, device=DEV)) for _ in range(4)]
    flat = torch.ones(4 * 64, device=DEV)
    views = [flat[i * 64:(i + 1) * 64] for i in range(4)]
    red = sr_dist.BucketedGradReducer(ps, views, [0, 64, 128, 192], flat, world=1, n_buckets=2, force=True,
                                      release="host")
    assert red.release == "host"
    red.arm()                              # epoch 1 announced, but no replay publishes it
    red.issue_all()                        # returns (does not raise): both buckets time out, are poisoned, and reduced
    torch.cuda.synchronize()
    lo = [b["lo"] for b in red.buckets]
    got = flat.cpu()
    assert all(torch.isnan(got[i]) for i in lo), got[lo]
    assert torch.isfinite(got).sum().item() == got.numel() - len(lo)
    with pytest.raises(RuntimeError, match=r"bucket\(s\) \[0, 1\]"):
        red.check()
    assert not red.status.any()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_overlap_modes_under_constrained_hardware_queues():
    """The device-released overlap spins a one-lane kernel at the head of the communication stream; that is only safe
    while the communication stream and the replaying stream own different hardware queues.  The
    reducer therefore probes the assumption when it is built and falls back to the host-released mode.  Here the whole
    overlapped iteration runs in a process of its own (HIP reads GPU_MAX_HW_QUEUES at start-up) with 1, 2 and 4
    hardware queues, and with each mode forced: every run must FINISH (no 120 s signal timeout — the subprocess limit is
    far below it), stay finite, report a mode consistent with its probe, and land on the same parameters bit for bit
    (one rank: the all-reduce is the identity, so the two modes and every queue count agree exactly)."""
    import json
    import subprocess
    import sys

    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "overlap_child.py")
    runs = []
    for hwq, forced in (("1", None), ("2", None), ("4", None), (None, "host"), (None, "device")):
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        env.pop("SR_GRAD_OVERLAP", None)
        env["SR_SIGNAL_TIMEOUT_S"] = "20"
        if hwq:
            env["GPU_MAX_HW_QUEUES"] = hwq
        if forced:
            env["SR_GRAD_OVERLAP"] = forced
        out = subprocess.run([sys.executable, child, str(_free_port())], env=env, capture_output=True, text=True,
                             timeout=150)
        assert out.returncode == 0, (hwq, forced, out.stdout[-2000:], out.stderr[-3000:])
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("CHILD ")][-1]
        r = json.loads(line[6:])
        print(hwq, forced, r["release"], r["probe"], r["overlap"]["bucket_done_ms_after_replay_end"])
        assert r["finite"]
        if forced:
            assert r["release"] == forced and r["probe"] is None
        else:
            assert r["release"] == ("device" if r["probe"]["independent"] else "host"), r
        runs.append(r)
    assert len({r["digest"] for r in runs}) == 1, [(r["hw_queues"], r["forced"], r["release"], r["digest"][:12]) for r in runs]


def test_timed_out_wait_poisons_the_bucket_and_the_guarded_step_is_refused(one_rank_group, monkeypatch):
    """after a wait gave up, the queued all-reduce ran on a half-written bucket and the optimiser graph was
    replayed on it.  Now the timed-out wait overwrites the bucket's first element with NaN (every rank receives it
    through the SUM), the guarded Adam step leaves parameters and moments untouched and raises its pinned word, and
    check() raises — also on a rank whose own waits were fine (simulated: NaN at a guard position without a status)."""
    from stylerenderer_amd import _lib
    from stylerenderer_amd import distributed as sr_dist
    from stylerenderer_amd.optim import FlatAdam

    monkeypatch.setenv("SR_SIGNAL_TIMEOUT_S", "0.2")
    ps = [torch.nn.Parameter(torch.randn(64, device=DEV)) for _ in range(4)]
    flat = torch.ones(4 * 64, device=DEV)
    offs = [0, 64, 128, 192]
    views = [flat[i * 64:(i + 1) * 64] for i in range(4)]
    red = sr_dist.BucketedGradReducer(ps, views, offs, flat, world=1, n_buckets=2, force=True, release="device")
    opt = red.guard(FlatAdam(ps, flat, lr=0.1, offs=offs))
    assert len(opt.guards) == 2 and list(opt.guards) == [b["lo"] for b in red.buckets]
    opt.step()                                         # a clean gradient: applied
    torch.cuda.synchronize()
    red.check()
    p1, m1, t1 = opt.flat_p.clone(), opt.m.clone(), float(opt.step_t)
    red.arm()                                          # epoch 1 announced, but nothing publishes it
    red._issue(1, replay=True)                         # wait (times out) + all-reduce of bucket 1 on the comm stream
    red.wait()
    opt.step()
    torch.cuda.synchronize()
    assert torch.isnan(flat[red.buckets[1]["lo"]]) and torch.isfinite(flat[:red.buckets[1]["lo"]]).all()
    assert torch.equal(opt.flat_p, p1) and torch.equal(opt.m, m1)          # the step was refused on the device
    assert float(opt.step_t) == t1                                         # ... and did not advance the step count
    with pytest.raises(RuntimeError, match=r"REFUSED.*bucket\(s\) \[1\] timed out"):
        red.check()
    red.check()                                        # reported once
    # a peer's marker (NaN arrives through the all-reduce, this rank's own status is clean)
    flat.fill_(1.0)
    flat[red.buckets[0]["lo"]] = float("nan")
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_p, p1)
    with pytest.raises(RuntimeError, match="REFUSED.*peer"):
        red.check()
    flat.fill_(1.0)
    opt.step()
    torch.cuda.synchronize()
    red.check()
    assert not torch.equal(opt.flat_p, p1)


def test_host_released_overlap_orders_buckets_before_the_replay_ends(one_rank_group, monkeypatch):
    """SR_GRAD_OVERLAP=host at full size: signal nodes store the epoch into pinned host words, issue_all() polls them
    and queues each collective when its bucket is complete — no kernel spins on the communication stream.  Same
    trajectory as the run without collectives, bit for bit; bucket 0 of the D phase is still reduced long before the
    replay ends."""
    monkeypatch.setenv("SR_GRAD_OVERLAP", "host")
    a, faces, data = full_size_trainer(force_collectives=True)
    monkeypatch.delenv("SR_GRAD_OVERLAP")
    b, _, _ = full_size_trainer()
    assert a.reduce_d.release == "host" and not a.reduce_d.counters.is_cuda and a.reduce_d.counters.is_pinned()
    batches = [data.batch(4) for _ in range(2)]
    for tr in (a, b):
        torch.manual_seed(123)
        tr.np_rng = np.random.RandomState(5)
        meshes = [tuple(t.clone() for t in faces.sample(4)) for _ in range(2)]
        torch.manual_seed(321)
        tr.logs = [tr.step(x, mesh=m) for x, m in zip(batches, meshes)]
    assert torch.equal(a.g_optim.flat_p, b.g_optim.flat_p) and torch.equal(a.d_optim.flat_p, b.d_optim.flat_p)
    assert a.logs == b.logs
    tries = [a.measure_overlap("d") for _ in range(3)]
    print("host-released overlap", tries)
    for t in tries:
        done = t["bucket_done_ms_after_replay_end"]
        assert len(done) == 4 and done == sorted(done) and t["release"] == "host"
    assert any(t["bucket_done_ms_after_replay_end"][0] < -0.25 * t["replay_ms"] for t in tries), tries
    assert a.reduce_d.counters.tolist() == [a.reduce_d.epoch] * 4


def test_gradients_written_into_the_flat_buffer_equal_the_copied_ones(monkeypatch):
    """distributed.BucketedGradReducer hands the convolution weights' slots of the flat gradient buffer to the kernel that
    writes their whole gradient (op.weight_prep._WPrepBwd): the parameters after three iterations (all four phases on the
    first) equal those of the copy-after-backward form bit for bit."""
    dev = torch.device("cuda")

    def run(inplace):
        monkeypatch.setenv("SR_GRAD_INPLACE", "1" if inplace else "0")
        faces = train.SyntheticFaceSource(dev, shape_dim=6, expression_dim=4, seed=3, face_sized=False)
        tr = graph_train.GraphedTrainer(size=32, latent=32, n_mlp=2, use_mesh=True, device=dev, seed=1, batch=4,
                                        mesh_vertices=faces.model.dim[2] // 3)
        data = train.SyntheticImages(8, 32, dev)
        for _ in range(3):
            tr.step(data.batch(4), faces=faces, log=False)
        torch.cuda.synchronize()
        claimed = len(tr.reduce_g.claimed) + len(tr.reduce_d.claimed)
        return tr.g_optim.flat_p.clone(), tr.d_optim.flat_p.clone(), claimed

    g1, d1, n1 = run(True)
    g0, d0, n0 = run(False)
    assert n0 == 0 and n1 > 10
    assert torch.equal(g1, g0) and torch.equal(d1, d0)
