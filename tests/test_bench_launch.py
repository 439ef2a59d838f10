"""CPU: `python bench.py --gpus 2` without a launcher must become two ranks (--gpus was parsed and
ignored).  Runs the same self-launch path on the gloo backend with a tiny model (`--plumbing`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing", "--steps", "2"] + list(extra),
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout            # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_self_launches_two_ranks_and_runs_the_train_leg():
    """Two gloo ranks run bench.py's OWN train_leg (GraphedTrainer capture=False, bucketed reducer) — the N > 1 branches
    of the line the driver's SCALE run parses: gradient_collective (bucket layout, issue order) and the all-reduce
    timing against the xGMI ring / direct-mesh bounds."""
    out = run_bench("--gpus", "2")
    assert out["n_gpus"] == 2 and out["plumbing"] is True and out["value"] > 0
    tl = out["train_step"]
    assert "error" not in tl and tl["value"] > 0 and tl["losses_finite"] is True and tl["parallelism"] == "dp2"
    assert tl["iters"] == 2 and tl["ms_per_iter"] > 0 and tl["host_enqueue_ms_per_iter"] > 0
    gc = tl["gradient_collective"]
    for k in ("generator_grads", "discriminator_grads"):
        assert gc[k]["mode"] == "overlapped, 4 buckets" and len(gc[k]["bucket_MB"]) == 4
        assert gc[k]["bucket_MB"] == sorted(gc[k]["bucket_MB"], reverse=True)          # the last bucket is the smallest
    assert set(gc) >= {"overlap_path_phase", "overlap_d_phase"}
    assert gc["eager_issue_order_path_phase"] == [0, 1, 2, 3]                           # in index order on every rank
    ar = tl["allreduce_125MB"]
    assert set(ar) >= {"bytes", "ms", "reduce_scatter_all_gather_ms", "ring_bound_ms", "direct_mesh_bound_ms",
                       "frac_of_ring_bound", "busbw_GBps"}
    assert ar["ms"] > 0 and ar["ring_bound_ms"] == ar["direct_mesh_bound_ms"] > 0      # equal at N = 2: 2(N-1)/N = 2/N


def test_train_leg_single_rank_has_no_collective_block():
    out = run_bench()
    tl = out["train_step"]
    assert tl["gradient_collective"] is None and "allreduce_125MB" not in tl and tl["parallelism"] == "dp1"


def test_single_rank_default():
    out = run_bench("--no-train")
    assert out["n_gpus"] == 1 and out["train_step"] is None


def test_world_size_mismatch_is_refused():
    env = {k: v for k, v in os.environ.items()}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing", "--gpus", "2"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_roofline_objects_prefer_counters_of_this_run():
    """bench.pmc_traffic / valu_roofline: counters collected by collect_live_pmc (this run's own rocprofv3 --pmc passes)
    win over the committed summary and are labelled as measured in the run; kernel names are shortened the way
    scripts/pmc_summary.py does it; 2*FETCH_SIZE + WRITE_SIZE (KB) per launch; VALU issue floor 4 cycles per wave64
    instruction over 1024 SIMDs at 2.4 GHz."""
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    assert bench._short_kernel("void (anonymous namespace)::k_grad_pix<float, 3, false>(long long, long long)") == \
        "k_grad_pix<float; 3; false>"
    committed = bench.pmc_traffic("k_conv_wino<8>")
    assert committed.get("traffic_measured_in_this_run") is False and committed["traffic"] > 0
    bench._LIVE_PMC.clear()
    bench._LIVE_PMC["k_conv_wino<8>"] = {"FETCH_SIZE": 100.0, "WRITE_SIZE": 50.0, "SQ_VALU_MFMA_BUSY_CYCLES": 768.0,
                                         "GRBM_GUI_ACTIVE": 8.0}
    try:
        live = bench.pmc_traffic("k_conv_wino<8>")
        assert live["traffic_measured_in_this_run"] is True and live["traffic"] == round(250.0 * 1024)
        assert live["mfma_pipe_busy"] == 0.75 and live["mfma_pipe_busy_measured_in_this_run"] is True
        bench._LIVE_PMC["a"] = {"SQ_INSTS_VALU": 3.0e7}
        bench._LIVE_PMC["b"] = {"SQ_INSTS_VALU": 1.0e7}
        v = bench.valu_roofline(["a", "b"], 0.1)
        assert v["counters_measured_in_this_run"] is True and v["wave_instructions_per_launch"] == 40000000
        assert abs(v["issue_floor_ms"] - 4.0e7 / (1024 * 2.4e9 / 4) * 1e3) < 1e-4 and abs(v["frac"] - v["issue_floor_ms"] / 0.1) < 1e-3
    finally:
        bench._LIVE_PMC.clear()
