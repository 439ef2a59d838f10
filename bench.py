#!/usr/bin/env python3
"""Benchmark of the StyleRenderer generator hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N > 1 and no WORLD_SIZE in the environment: re-launches itself as N ranks under
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (driver form)

Headline metric (BASELINE.json): generator fwd+bwd images/s at 256x256.  Workload = BASELINE config[1]:
`Generator(256, 512, 8, channel_multiplier=2)` forward + backward, batch 16 per GPU, fresh random
latents every step, fresh per-layer noise, loss = image.sum(), fp32 end to end (exact-fp32 MFMA).
Inputs are generated on the device (resident in HBM when the timed region starts).
One process per GPU; N > 1 is data parallel (weak scaling: 16 images per GPU) with the gradient
all-reduce over RCCL overlapped with backward (nccl backend == RCCL on ROCm).

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  roofline      achieved TFLOP/s of the dominant kernel (3x3 stride-1 MFMA convolution, forward and
                data-gradient launches) measured with events on the launch stream INSIDE the timed
                steps, against the 157.3 TFLOP/s fp32-MFMA peak of gfx950 (frac = EXECUTED MFMA flops);
  cpu_baseline  the CPU oracle (oracle/model_oracle.py, a restatement of the reference's grouped-
                convolution formulation) timed on this host's cores on a bounded sample;
  train_step    BASELINE config[2]: the full G+D iteration (GeneratorWithMap(256) + Discriminator(256),
                4 images per GPU, synthetic images + 3DMM-size mesh, lazy R1 / path-length cadence 16 / 4,
                64 timed iterations), images/s, enqueue time, and for N > 1 the measured 125 MB gradient
                all-reduce against the xGMI ring bound;
  rasterizer    BASELINE config[3]: Mtri/s forward and forward+backward, its own HBM roofline, the
                vector-ALU issue roofline that actually bounds it (roofline_valu, from the committed PMC pass) and
                the single-thread C oracle (the reference's CPU rasterizer restated) beside it;
  inversion     BASELINE config[4]: 400-step latent inversion (generator + rasterizer + LPIPS metric), steps/s, with a
                roofline object (executed MFMA flops, launches and microseconds per launch of the replayed step);
  step_executed_mfma_frac   executed matrix-core flops of the WHOLE headline step / wall time / 157.3 TFLOP/s.
`roofline.traffic`, `mfma_pipe_busy` and the rasterizer's traffic / VALU instruction counts are measured IN this run:
after the timed steps rank 0 spawns three short `rocprofv3 --pmc` passes (counters only, one TCC counter per pass) over
a child of this script (`collect_live_pmc`; `--no-pmc` skips them and cites the committed profiles/ summary instead).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBPS = 8000.0                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s
XGMI_LINK_GBPS = 153.0                 # per link and direction; 7 links per GPU
FLOP_PER_IMAGE_FWD_BWD = 270.7e9       # 3 x 90.24 GFLOP of convolution per image
G_PARAM_BYTES = 31.29e6 * 4            # Generator(256) gradients per step


def executed_flops(kind, geom, flops):
    """Matrix-core flops a convolution / weight-gradient launch EXECUTES for `flops` direct-convolution flops: the
    Winograd-eligible stride-1 3x3 shapes (op/conv.py's dispatch rule) issue 16/36 of the direct multiplies."""
    k, stride, transposed, _b, c, n, gh, gw = geom
    wino = (k == 3 and stride == 1 and transposed == 0 and c % 64 == 0 and n % 64 == 0 and
            (gw % 32 == 0 and gh % 8 == 0 if kind == "conv" else gw % 16 == 0 and gh % 2 == 0))
    return flops * (16.0 / 36.0 if wino and os.environ.get("SR_WINOGRAD", "1") != "0" else 1.0)


_STASH = {"line": None}


def _fail_fast(rank, world, what, exc_text):
    """A leg that raised on THIS rank of a multi-rank run: the other ranks are (or will be) blocked in collectives
    this rank will never enter.  Print what is already measured (rank 0) and leave at once — the launcher tears the
    job down — instead of hanging every rank until an outer time limit."""
    sys.stderr.write("bench.py: rank %d failed in %s; aborting the %d-rank job\n%s\n" % (rank, what, world, exc_text))
    sys.stderr.flush()
    if rank == 0 and _STASH["line"] is not None:
        line = dict(_STASH["line"])
        line[what] = {"error": exc_text.strip().splitlines()[-1][:400]}
        print(json.dumps(line), flush=True)
    os._exit(13)


def _on_sigterm(signum, frame):
    # a PEER rank failed and the launcher is tearing the job down: rank 0 still reports the headline it holds
    if _STASH["line"] is not None:
        line = dict(_STASH["line"])
        line["aborted"] = "peer rank failed (see stderr)"
        print(json.dumps(line), flush=True)
    os._exit(14)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (generator leg)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-raster", action="store_true", help="skip the rasterizer (BASELINE config[3]) leg")
    ap.add_argument("--no-train", action="store_true", help="skip the full G+D step (BASELINE config[2]) leg")
    ap.add_argument("--no-inversion", action="store_true", help="skip the latent-inversion (BASELINE config[4]) leg")
    ap.add_argument("--inversion-steps", type=int, default=400)
    ap.add_argument("--train-iters", type=int, default=64, help="timed iterations of the G+D step leg")
    ap.add_argument("--train-batch", type=int, default=4, help="images per GPU of the G+D step leg")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--cpu-threads", default="32,64,128",
                    help="thread counts of the CPU baseline's sweep (comma separated); the best one runs the sample")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not spawn the rocprofv3 --pmc passes that measure HBM traffic / MFMA busy / VALU "
                         "instructions for the roofline objects (they then cite the committed profiles/ summary)")
    ap.add_argument("--no-split-bf16", action="store_true",
                    help="skip the secondary timing of the step with SR_CONV_SPLIT_BF16=1 (opt-in split-bf16 weight gradients)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--step-traffic", nargs="?", const="1", default=None, metavar="JSON",
                    help="collect the whole-step HBM traffic of the config[2] / config[4] legs live (minutes of "
                         "rocprofv3 --pmc passes) and, with a path, write the summary there (profiles/rNN_step_traffic.json)")
    ap.add_argument("--plumbing", action="store_true",
                    help="CPU-only launch check (gloo, tiny model): exercises --gpus N rank spawning without a GPU")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become N ranks (one per GPU) on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------
def raster_leg(dev, world, batch=64, res=256, iters=10, cpu_baseline=True):
    """Second half of the metric: rasterizer Mtri/s on BASELINE config[3] (BFM-size-class mesh,
    ~50k triangles, 256x256, batch 64 per GPU, int64 ids as the API demands), forward and forward+backward
    (fused attribute interpolation + deterministic gradient gather).  Algorithmic HBM bytes per image
    24*nf + 12*nv + 36*h*w forward, (24+12+4c)*h*w + 24*nv backward."""
    import importlib

    import torch

    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth

    rz = importlib.import_module("stylerenderer_amd.op.rasterize")      # (op.rasterize is the function)

    v0, tri = synth.face_sized_mesh()
    vh = synth.random_poses(v0, batch, seed=1234)
    v = torch.from_numpy(vh).to(dev)
    nrm = torch.from_numpy(synth.vertex_normals(vh, tri)).to(dev)
    t = torch.from_numpy(tri).to(dev)
    nf, nv = tri.shape[0], v0.shape[0]

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                      # the raster kernels are launched on torch's current stream
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    ms_api = timed(lambda: rz.forward(v, t, res, res, False, 1e-6))      # index + coeff written (API parity)
    ms_f = timed(lambda: op.rasterize(v, nrm, t, res))
    vg, ng = v.clone().requires_grad_(), nrm.clone().requires_grad_()

    gout = torch.randn(batch, res, res, 3, device=dev)         # a dense upstream gradient, resident like the inputs

    def fb():
        vg.grad = ng.grad = None
        op.rasterize(vg, ng, t, res).backward(gout)

    ms_fb = timed(fb)
    # the gradient pass alone: backward of a recorded forward (sr_rasterize_grad_f32 and its torch glue), timed with
    # events like the forward — not the difference of two timings
    held = op.rasterize(vg, ng, t, res)

    def bwd_only():
        vg.grad = ng.grad = None
        held.backward(gout, retain_graph=True)

    ms_b = timed(bwd_only)
    c = 3
    fwd_bytes = batch * (24 * nf + 12 * nv + 36 * res * res)
    bwd_bytes = batch * ((24 + 12 + 4 * c) * res * res + 24 * nv)
    out = {"workload": "BASELINE config[3]: nv=%d nf=%d, %dx%d, batch %d per GPU" % (nv, nf, res, res, batch),
           "fwd_mtri_s": round(world * batch * nf / ms_f / 1e3, 1), "fwd_ms": round(ms_f, 4),
           "fwd_api_ms": round(ms_api, 4),
           "fwd_bwd_mtri_s": round(world * batch * nf / ms_fb / 1e3, 1), "fwd_bwd_ms": round(ms_fb, 4),
           "bwd_ms": round(ms_b, 4),
           "roofline": {"bound": "hbm", "kernel": "sr_rasterize_forward_f32 (k_tile_zero + k_tile_bin_lds + k_tile_raster: "
                                                  "LDS-staged 32x32 triangle tiles)",
                        "achieved": round(fwd_bytes / ms_api / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(fwd_bytes / ms_api / 1e6 / HBM_PEAK_GBPS, 4), "traffic": None,
                        "bytes_per_launch": fwd_bytes,
                        "note": "per-triangle setup / per-pixel shading (vector ALU) bound, priced against the HBM roof as "
                                "SURVEY 8(d) asks; roofline_valu prices it against the vector-ALU issue rate"},
           "roofline_bwd": {"bound": "hbm", "kernel": "sr_rasterize_grad_f32 (k_grad_big + k_grad_pix + k_grad_vert; the "
                                                      "leader table comes from the tiled forward), timed as the backward "
                                                      "of a recorded forward",
                            "achieved": round(bwd_bytes / ms_b / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(bwd_bytes / ms_b / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_launch": bwd_bytes},
           "bit_exact_vs_cpu_oracle": "tests/test_ops_gpu.py"}
    fwd_kernels = ["k_tile_zero", "k_tile_bin_lds<false>", "k_tile_raster<false>"]
    bwd_kernels = ["k_grad_big<float; 3; false>", "k_grad_pix<float; 3; false>", "k_grad_vert<float; 3; false>"]
    out["roofline"].update(pmc_traffic(fwd_kernels))
    out["roofline_bwd"].update(pmc_traffic(bwd_kernels))
    # the bound these kernels actually run against: vector-ALU issue.  A wave64 VALU instruction occupies its SIMD for 4
    # cycles, so a launch that issues I wave-instructions needs at least 4 I / (1024 SIMDs x 2.4 GHz); I = SQ_INSTS_VALU
    # of the committed PMC pass (profiles/r*_raster_pmc.csv, scripts/raster_pmc.sh)
    out["roofline_valu"] = valu_roofline(fwd_kernels, ms_api)
    out["roofline_valu_bwd"] = valu_roofline(bwd_kernels, ms_b)
    if cpu_baseline:
        import raster as oracle_raster

        nb = 8
        oracle_raster.forward_buffers(vh[:1], tri, res, res, False, 1e-6)              # build + warm
        t0 = time.perf_counter()
        reps = 0
        while reps < 3 or time.perf_counter() - t0 < 2.0:
            oracle_raster.forward_buffers(vh[:nb], tri, res, res, False, 1e-6)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": round(nb * nf / dt / 1e6, 2), "unit": "Mtri/s", "cores": 1, "kind": "port",
                               "sample": "oracle/rasterize_oracle.c (sequential CPU loops of reference "
                                         "op/rasterize.cpp:21-67), forward, batch %d of the same mesh, %d repeats, "
                                         "single thread like the reference" % (nb, reps)}
    return out


_LIVE_PMC = {}         # kernel -> {counter: mean per dispatch}, filled by collect_live_pmc() in this run


def _short_kernel(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(",", ";")


def collect_live_pmc(timeout_s=150):
    """Hardware counters of THIS run's kernels: three short rocprofv3 --pmc passes (counters only — no trace domains —
    one TCC counter per pass, as MI355X_MICROARCH.md prescribes) over a child of this very script (2 generator steps +
    the rasterizer leg).  Fills _LIVE_PMC; returns a note for the JSON line (None when everything worked).  The timed
    measurements of the parent are finished or not yet started when this runs: nothing overlaps with them."""
    import csv
    import glob
    import shutil
    import tempfile

    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return "this run is itself being profiled: no nested --pmc passes"
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="sr_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-train", "--no-inversion",
             "--no-cpu-baseline", "--no-split-bf16", "--pmc-child"]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    note = None
    passes = (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"])
    acc = {}
    for i, counters in enumerate(passes):
        d = os.path.join(out, "p%d" % i)
        try:
            subprocess.run([tool, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                           cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception as e:           # noqa: BLE001  (a profiler problem must not take the benchmark down)
            note = "pmc pass %s failed: %s" % (counters[0], type(e).__name__)
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            note = "pmc pass %s wrote no counters" % counters[0]
            continue
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                a = acc.setdefault((_short_kernel(r["Kernel_Name"]), r["Counter_Name"]), [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for (k, c), (v, n) in acc.items():
        _LIVE_PMC.setdefault(k, {})[c] = v / n
    shutil.rmtree(out, ignore_errors=True)
    return note


def graph_step_traffic(probe, probe_args, segments, timeout_s=420):
    """HBM bytes of hipGraph-replayed work (the config[2] phases, the config[4] inversion step): two rocprofv3 --pmc passes
    (FETCH_SIZE, then WRITE_SIZE — counters only, one TCC counter per pass) over scripts/<probe>, whose LAST kernel
    dispatches are the replays in question.  `segments` = [(name, dispatches), ...] in execution order: the tail of the
    process's dispatch list is cut into those pieces and every piece's counters are summed over ALL its kernels
    (2 * FETCH_SIZE + WRITE_SIZE KB: the gfx950 correction of pmc_traffic).  Returns ({name: bytes} or None, note).
    Counter collection costs ~10 ms per dispatch: this is run by `bench.py --step-traffic` (scripts/profile_round.sh),
    not by the default run, which reads the committed summary profiles/r*_step_traffic.json."""
    import csv
    import glob
    import shutil
    import tempfile

    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled: no nested --pmc passes"
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    tail = sum(n for _k, n in segments)
    if tool is None or not tail:
        return None, "rocprofv3 not found" if tool is None else "no dispatch count"
    out = tempfile.mkdtemp(prefix="sr_pmc_step_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    child = [sys.executable, os.path.join(ROOT, "scripts", probe)] + [str(a) for a in probe_args]
    sums, note = {}, None
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, c)
        try:
            subprocess.run([tool, "--pmc", c, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child, cwd="/tmp",
                           env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception as e:           # noqa: BLE001
            note = "pmc pass %s failed: %s" % (c, type(e).__name__)
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            note = "pmc pass %s wrote no counters" % c
            continue
        with open(files[0]) as f:
            rows = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(f) if r["Counter_Name"] == c]
        rows.sort()
        if len(rows) < tail:
            note = "pmc pass %s saw %d dispatches, fewer than the %d of the replays" % (c, len(rows), tail)
            continue
        vals = [v for _i, v in rows[-int(tail):]]
        per, pos = {}, 0
        for name, n in segments:
            per[name] = sum(vals[pos:pos + n])
            pos += n
        sums[c] = per
    shutil.rmtree(out, ignore_errors=True)
    if len(sums) != 2:
        return None, note
    return {k: (2.0 * sums["FETCH_SIZE"][k] + sums["WRITE_SIZE"][k]) * 1024.0 for k, _n in segments}, None


def committed_step_traffic():
    """The newest profiles/r*_step_traffic.json (written by `bench.py --step-traffic` in scripts/profile_round.sh)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f), os.path.basename(files[-1])
    except Exception:                    # noqa: BLE001
        return None, None


VALU_PEAK_GINST = 1024 * 2.4 / 4.0      # wave64 VALU instructions / ns over 1024 SIMDs at 2.4 GHz, 4 cycles each


def valu_roofline(kernel_rows, measured_ms):
    """Vector-ALU issue bound of an operator made of `kernel_rows`: SQ_INSTS_VALU (wave-instructions per launch, from
    the newest committed profiles/r*_raster_pmc.csv) against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction."""
    import csv
    import glob

    live = all(k in _LIVE_PMC and "SQ_INSTS_VALU" in _LIVE_PMC[k] for k in kernel_rows)
    insts, seen, files = 0.0, 0, []
    if live:
        insts, seen = sum(_LIVE_PMC[k]["SQ_INSTS_VALU"] for k in kernel_rows), len(kernel_rows)
    else:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_raster_pmc.csv")))
        if not files:
            return None
        with open(files[-1]) as f:
            for r in csv.DictReader(l for l in f if not l.startswith("#")):
                if r["kernel"] in kernel_rows and r.get("SQ_INSTS_VALU"):
                    insts += float(r["SQ_INSTS_VALU"])
                    seen += 1
    if seen != len(kernel_rows) or not measured_ms:
        return None
    floor_ms = insts / (VALU_PEAK_GINST * 1e9) * 1e3
    return {"bound": "valu", "wave_instructions_per_launch": round(insts), "peak": round(VALU_PEAK_GINST, 1),
            "unit": "G wave-inst/s", "achieved": round(insts / (measured_ms * 1e-3) / 1e9, 1),
            "frac": round(floor_ms / measured_ms, 4), "issue_floor_ms": round(floor_ms, 4),
            "counters_measured_in_this_run": bool(live),
            "source": ("rocprofv3 --pmc SQ_INSTS_VALU pass spawned by this run (collect_live_pmc)" if live else
                       "profiles/%s (SQ_INSTS_VALU, separate rocprofv3 --pmc pass)" % os.path.basename(files[-1]))}


def pmc_traffic(kernel_rows):
    """HBM bytes per launch of `kernel_rows` (one name, or several kernels of one operator: summed) from the newest
    committed PMC summary (profiles/r*_pmc.csv, written by scripts/profile_round.sh + scripts/pmc_summary.py from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command).  Units are KB; gfx950's
    FETCH_SIZE reports half of a 16 B/lane streaming read (MI355X_MICROARCH.md, HBM section; confirmed here on
    k_nba_bwd and k_rowdot, whose read:write byte ratios are known), hence 2*FETCH + WRITE."""
    import csv
    import glob

    if isinstance(kernel_rows, str):
        kernel_rows = [kernel_rows]
    if all(k in _LIVE_PMC and "FETCH_SIZE" in _LIVE_PMC[k] and "WRITE_SIZE" in _LIVE_PMC[k] for k in kernel_rows):
        # counters of THIS run (collect_live_pmc: separate --pmc passes over a child of this script)
        total = sum((2.0 * _LIVE_PMC[k]["FETCH_SIZE"] + _LIVE_PMC[k]["WRITE_SIZE"]) * 1024.0 for k in kernel_rows)
        out = {"traffic": round(total), "traffic_unit": "bytes/launch", "traffic_measured_in_this_run": True,
               "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes spawned by this run "
                                 "(2*FETCH_SIZE+WRITE_SIZE, KB)"}
        c = _LIVE_PMC[kernel_rows[0]]
        if len(kernel_rows) == 1 and c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            out["mfma_pipe_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 3)
            out["mfma_pipe_busy_measured_in_this_run"] = True
        return out
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.csv")) if "raster_pmc" not in f)
    if not files:
        return {}
    total, seen, busy = 0.0, 0, None
    with open(files[-1]) as f:
        for r in csv.DictReader(l for l in f if not l.startswith("#")):
            if r["kernel"] in kernel_rows and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
                total += (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0
                seen += 1
                if r.get("SQ_VALU_MFMA_BUSY_CYCLES") and r.get("GRBM_GUI_ACTIVE") and float(r["GRBM_GUI_ACTIVE"]) > 0:
                    busy = float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / (float(r["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0)
    if seen != len(kernel_rows):
        return {}
    # PMC counters need their own rocprofv3 passes: this value is READ from the committed summary of the builder's
    # profile of this same command, it is not collected inside this run
    out = {"traffic": round(total), "traffic_unit": "bytes/launch", "traffic_measured_in_this_run": False,
           "traffic_source": "profiles/%s (2*FETCH_SIZE+WRITE_SIZE)" % os.path.basename(files[-1])}
    if len(kernel_rows) == 1 and busy is not None:
        # fraction of the kernel's cycles in which the matrix pipe was busy (same PMC file, separate pass):
        # SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs — independent of the clock the
        # roof is quoted at (under these kernels the shader clock is 1.8-2.1 GHz)
        out["mfma_pipe_busy"] = round(busy, 3)
        out["mfma_pipe_busy_measured_in_this_run"] = False
    return out


# ---------------------------------------------------------------------------------------------------
def _time_ms(fn, reps, on_gpu):
    """Milliseconds per call of `fn` (a collective), after 3 untimed calls; device events on a GPU, wall clock on CPU."""
    import torch

    for _ in range(3):
        fn()
    if on_gpu:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def train_leg(dev, rank, world, iters, batch, size=256, plumbing=False):
    """BASELINE config[2]: the reference's full training iteration (train.py:239-358) — D step, lazy R1
    every 16, G step, lazy path-length regulariser every 4 on batch // 2 (double backward through every
    operator), EMA — on GeneratorWithMap + Discriminator, `batch` images per GPU (4 x 8 GPUs = global 32),
    synthetic in-memory images and a 3DMM-size-class mesh sampled per step.  The timed
    window starts at iteration 0 of the lazy-regulariser cadence, so `iters` = 64 holds 4 R1 and 16
    path-length iterations like any aligned window of a long run.
    plumbing=True (bench.py --plumbing, tests/test_bench_launch.py): the SAME function on CPU tensors over gloo —
    GraphedTrainer(capture=False) on an 8x8 network and a small mesh: the phases, bucket hooks, collectives and every
    N > 1 branch of this leg (gradient_collective, allreduce timing against the xGMI bounds) run without a GPU."""
    import torch
    import torch.distributed as dist

    from stylerenderer_amd import graph_train, train

    on_gpu = dev.type == "cuda"
    latent, n_mlp = (32, 2) if plumbing else (512, 8)
    faces = train.SyntheticFaceSource(dev, seed=0, face_sized=not plumbing)
    graphs = os.environ.get("SR_TRAIN_GRAPHS", "1") != "0"
    if graphs:
        # forward + backward of each phase and the Adam steps replayed from hipGraphs (graph_train.py)
        tr = graph_train.GraphedTrainer(size=size, latent=latent, n_mlp=n_mlp, channel_multiplier=2, use_mesh=True,
                                        device=dev, seed=0, batch=batch, mesh_vertices=faces.model.dim[2] // 3,
                                        capture=on_gpu)
    else:
        tr = train.Trainer(size=size, latent=latent, n_mlp=n_mlp, channel_multiplier=2, use_mesh=True, device=dev, seed=0)
    data = train.SyntheticImages(64, size, dev)

    def fence():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(2):                                   # iterations 0 (both regularisers) and 1 (plain)
        tr.step(data.batch(batch), faces=faces, log=False)
    tr.iteration = 0
    fence()
    t0 = time.perf_counter()
    last = None
    for _ in range(iters):
        last = tr.step(data.batch(batch), faces=faces, log=False)
    fence()
    elapsed = time.perf_counter() - t0
    # host cost of enqueueing ONE iteration on an idle GPU (inside the long loop the host is throttled — a graph is not
    # re-launched while its previous launch runs — so its loop time equals the GPU time whatever bounds the step)
    enq = []
    for _ in range(3):
        fence()
        t1 = time.perf_counter()
        last = tr.step(data.batch(batch), faces=faces, log=False)
        enq.append(time.perf_counter() - t1)
    fence()
    t_enq = sorted(enq)[1] * iters
    if world > 1:
        t = torch.tensor([elapsed, t_enq], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_enq = float(t[0]), float(t[1])
    finite = all(bool(torch.isfinite(v)) for v in last.values())
    # ---- where the iteration goes: device time per phase (timed replays) and its MFMA-convolution work, counted
    # by running every phase body once eagerly with the per-launch hook of op.conv (algorithmic = direct-convolution
    # flops 2*B*H*W*Cin*Cout*k^2 of every convolution / weight-gradient launch; executed = what the matrix cores
    # issue: 16/36 of that on the Winograd-eligible stride-1 3x3 launches)
    phases, roof = None, None
    if graphs and on_gpu:
        # EVERY rank runs this block: the phases issue their gradient collectives (replayed: after the replay; eager:
        # from the backward hooks), a rank that sat it out would leave the others' all-reduces unmatched.  Rank 0 reports.
        from stylerenderer_amd.op import conv as conv_op

        cadence = {"d": 1.0, "r1": 1.0 / 16, "g": 1.0, "path": 1.0 / 4, "d_opt": 1.0 + 1.0 / 16, "g_opt": 1.0 + 1.0 / 4,
                   "ema": 1.0}
        phases = {}
        for name in ("d", "r1", "g", "path", "d_opt", "g_opt", "ema"):
            phases[name] = {"ms_per_replay": round(tr.time_phase(name, 3), 3), "per_iteration": round(cadence[name], 4),
                            "kernel_nodes": getattr(tr.graphs[name], "kernel_nodes", None)}
        for name in ("d", "r1", "g", "path"):
            conv_op.PROFILE = []
            tr._eager_phase(name)
            torch.cuda.synchronize()
            prof, conv_op.PROFILE = conv_op.PROFILE, None
            alg = sum(fl for (_k, _g, fl, _a, _b) in prof)
            exe = sum(executed_flops(kind, geom, fl) for kind, geom, fl, _a, _b in prof)
            ms = phases[name]["ms_per_replay"]
            phases[name].update({"mfma_launches": len(prof), "algorithmic_gflop": round(alg / 1e9, 1),
                                 "executed_gflop": round(exe / 1e9, 1),
                                 "executed_tflops": round(exe / (ms * 1e-3) / 1e12, 1),
                                 "frac_of_fp32_mfma_peak": round(exe / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3)})
        alg_it = sum(phases[n]["algorithmic_gflop"] * cadence[n] for n in ("d", "r1", "g", "path"))
        exe_it = sum(phases[n]["executed_gflop"] * cadence[n] for n in ("d", "r1", "g", "path"))
        ms_it = elapsed / iters * 1e3
        roof = {"bound": "mfma", "kernel": "all MFMA convolution / weight-gradient launches of an average iteration "
                                          "(k_conv_wino, k_wgrad_wino, k_convt_fused, k_conv_mfma, k_wgrad_mfma)",
                "achieved": round(exe_it / ms_it, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(exe_it / ms_it / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "executed_gflop_per_iteration": round(exe_it, 1), "algorithmic_gflop_per_iteration": round(alg_it, 1),
                "kernel_launches_per_iteration": round(sum((phases[n].get("kernel_nodes") or 0) * cadence[n]
                                                           for n in phases), 1),
                "algorithmic_tflops": round(alg_it / ms_it, 2),
                "note": "whole-iteration rate: executed matrix-core flops of every phase at its cadence / measured "
                        "ms per iteration (includes all non-MFMA kernels and the optimiser steps in the time)"}
        if rank != 0:
            phases, roof = None, None
    collective = None
    if graphs and world > 1:
        # chosen collective, bucket sizes, and when each bucket's reduction finished relative to the END of the
        # path-length backward (negative = overlapped with it)
        collective = {"generator_grads": tr.reduce_g.describe(), "discriminator_grads": tr.reduce_d.describe(),
                      "overlap_path_phase": tr.measure_overlap("path"), "overlap_d_phase": tr.measure_overlap("d")}
        if not on_gpu:
            # eager (CPU / gloo) form of the same measurement: the order in which the hooks issued the buckets of the
            # last path-length phase and how many were on the wire before the backward produced its last gradient
            tr._eager_phase("path")
            issues = [b for kind, b, _t in tr.reduce_g.log if kind == "issue"]
            collective["eager_issue_order_path_phase"] = issues
    out = {"workload": "BASELINE config[2]: GeneratorWithMap(%d) + Discriminator(%d) full G+D step, %d img/GPU "
                       "(global %d), d_reg_every 16, g_reg_every 4, path batch %d, synthetic images + mesh "
                       "nv=%d nf=%d" % (size, size, batch, batch * world, max(1, batch // 2),
                                        faces.model.dim[2] // 3, faces.tri.shape[0]),
           "value": round(batch * world * iters / elapsed, 2), "unit": "images/s", "iters": iters,
           "ms_per_iter": round(elapsed / iters * 1e3, 3),
           "host_enqueue_ms_per_iter": round(t_enq / iters * 1e3, 3),
           "launch_bound": bool(t_enq > 0.95 * elapsed),
           "execution": ("hipGraph replay per phase (graph_train.GraphedTrainer); bucketed all-reduce of the flat "
                         "gradient buffer on a communication stream: each bucket is released by a signal kernel node "
                         "inside the replayed backward (sr_signal_set publishes the replay's epoch) that a one-lane "
                         "polling kernel in front of the collective waits for (sr_signal_wait_timeout)"
                         if graphs and on_gpu else
                         "graph_train.GraphedTrainer(capture=False): the same phases, bucket hooks and collectives "
                         "launched eagerly" if graphs else "eager launches (train.Trainer, DDP buckets)"),
           "losses_finite": finite, "parallelism": "dp%d" % world, "gradient_collective": collective,
           "roofline": roof, "phases": phases}
    if world > 1:
        nbytes = G_PARAM_BYTES if not plumbing else 1.0e6        # (the CPU stand-in keeps the keys, not the size)
        n = int(nbytes // 4)
        buf = torch.zeros(n, device=dev)
        ms = _time_ms(lambda: dist.all_reduce(buf), 10, on_gpu)
        # the same reduction as reduce-scatter + all-gather (each GPU owns 1/N of the buffer in between): on the
        # fully connected xGMI mesh every peer link carries 1/N of the data at once instead of a ring's hops
        n_pad = (n + world - 1) // world * world
        full = torch.zeros(n_pad, device=dev)
        shard = torch.zeros(n_pad // world, device=dev)

        def rsag():
            dist.reduce_scatter_tensor(shard, full)
            dist.all_gather_into_tensor(full, shard)

        try:
            ms_rsag = round(_time_ms(rsag, 10, on_gpu), 3)
        except RuntimeError:             # a backend without reduce_scatter_tensor (gloo): every rank lands here
            ms_rsag = None
        ring = 2.0 * (world - 1) / world * nbytes / (XGMI_LINK_GBPS * 1e9) * 1e3
        direct = 2.0 * (nbytes / world) / (XGMI_LINK_GBPS * 1e9) * 1e3
        out["allreduce_125MB"] = {"bytes": int(nbytes), "ms": round(ms, 3), "reduce_scatter_all_gather_ms": ms_rsag,
                                  "ring_bound_ms": round(ring, 3),
                                  "direct_mesh_bound_ms": round(direct, 3),
                                  "frac_of_ring_bound": round(ring / ms, 3) if ms > 0 else None,
                                  "busbw_GBps": round(2.0 * (world - 1) / world * nbytes / ms / 1e6, 1)}
    del tr, data, faces
    if on_gpu:
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------
def inversion_leg(dev, steps=400, size=256):
    """BASELINE config[4]: 400 Adam steps over the W+ latent and the mesh pose through GeneratorWithMap(256), the
    rasterizer (forward + deterministic backward) and the LPIPS VGG16 metric (lpips.PNetLin, reference heads), one hipGraph per iteration
    (inversion.LatentInverter).  Random-init generator / metric weights (no checkpoints offline); the target is a
    rendering of a different latent and pose."""
    import torch

    from stylerenderer_amd import inversion, lpips, model, synth

    torch.manual_seed(0)
    g = model.GeneratorWithMap(size, 512, 8, channel_multiplier=2).to(dev)
    net = lpips.PNetLin().to(dev)
    v0, tri = synth.face_sized_mesh()
    v = torch.from_numpy(v0[None]).to(dev)
    nrm = torch.from_numpy(synth.vertex_normals(v0[None], tri)).to(dev)
    mesh = (v, nrm, torch.from_numpy(tri).to(dev))
    with torch.no_grad():
        w_true = g.style(torch.randn(1, 512, device=dev)).unsqueeze(1).repeat(1, g.n_latent, 1)
        rot = inversion.utils_3d.euler_mat(torch.tensor([[0.3, -0.1, 0.05]], device=dev), "yxz")[0]
        posed = (torch.matmul(v, rot).contiguous(), torch.matmul(nrm, rot).contiguous(), mesh[2])
        noise = [n.detach() for n in g.make_noise()]
        target, _, _ = g([w_true], posed, input_is_latent=True, noise=noise)

    from stylerenderer_amd.op import conv as conv_op

    census = {}

    def timed(use_graph, n):
        inv = inversion.LatentInverter(g, net, target, mesh, noise=noise, use_graph=use_graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = inv.run(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        h = hist.cpu()
        steady = None
        if use_graph:
            census["kernel_nodes"] = getattr(inv.graph, "kernel_nodes", None)
            # replay rate alone (the 400-step figure above includes mean-latent draws, 3 eager warm-up iterations
            # and the capture)
            t1 = time.perf_counter()
            inv.run(100)
            torch.cuda.synchronize()
            steady = 100 / (time.perf_counter() - t1)
            # matrix-core work of ONE step (every convolution / weight-gradient launch of generator + VGG trunk,
            # forward and backward), counted by running the same iteration once eagerly under op.conv's launch hook
            conv_op.PROFILE = []
            inv._iteration()
            torch.cuda.synchronize()
            prof, conv_op.PROFILE = conv_op.PROFILE, None
            census["mfma_launches"] = len(prof)
            census["alg"] = sum(fl for (_k, _g, fl, _a, _b) in prof)
            census["exe"] = sum(executed_flops(kind, geom, fl) for (kind, geom, fl, _a, _b) in prof)
        return n / dt, float(h[0]), float(h[-1]), bool(torch.isfinite(h).all()), steady

    # one short untimed inversion first, like the W warm-up steps of the headline: the caching allocator and the
    # per-weight caches of the frozen networks (prepared / Winograd-domain weights, adjoints: they live on the NETWORK and
    # serve every later inverter) are cold exactly once per process.  The timed run below still includes its own eager
    # warm-up iterations and the graph capture.
    inversion.LatentInverter(g, net, target, mesh, noise=noise, use_graph=False).run(2)
    torch.cuda.synchronize()
    sps, l0, l1, finite, steady = timed(True, steps)
    sps_eager = timed(False, max(8, steps // 10))[0]
    del g, net
    torch.cuda.empty_cache()
    roof = None
    if steady and census.get("exe"):
        ms = 1e3 / steady
        nodes = census.get("kernel_nodes") or 0
        roof = {"bound": "mfma", "kernel": "all MFMA convolution / weight-gradient launches of one replayed step "
                                          "(GeneratorWithMap(256) + VGG16 trunk, forward + backward, batch 1)",
                "achieved": round(census["exe"] / ms / 1e9, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(census["exe"] / ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "executed_gflop_per_step": round(census["exe"] / 1e9, 1),
                "algorithmic_gflop_per_step": round(census["alg"] / 1e9, 1),
                "ms_per_replay": round(ms, 3), "kernel_launches_per_step": nodes,
                "mfma_launches_per_step": census.get("mfma_launches"),
                "avg_us_per_launch": round(ms * 1e3 / nodes, 2) if nodes else None,
                "mfma_time_at_peak_ms": round(census["exe"] / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3, 3),
                "note": "batch 1: the step is bounded by launch count x per-launch latency inside the graph and by "
                        "tiles that cannot fill 256 CUs (a 4x4..32x32 map at batch 1 is 1-16 workgroups), not by a "
                        "roof: mfma_time_at_peak_ms is what the matrix-core work alone would take"}
    return {"roofline": roof, "workload": "BASELINE config[4]: latent inversion, %d Adam steps, GeneratorWithMap(%d) + rasterizer "
                        "(nv=%d nf=%d) + LPIPS-shaped VGG16 metric, batch 1" % (steps, size, v0.shape[0], tri.shape[0]),
            "value": round(sps, 2), "unit": "steps/s", "steps": steps,
            "seconds_for_%d_steps" % steps: round(steps / sps, 2),
            "eager_steps_per_s": round(sps_eager, 2), "replay_steps_per_s": round(steady, 2),
            "execution": "one hipGraph replay per step (capture included in the timed run; replay_steps_per_s = 100 "
            "further replays alone); frozen networks prepare their weights once (op.weight_prep) — one untimed 2-step "
            "inversion warms the allocator and those per-network caches first", "loss_first": round(l0, 5), "loss_last": round(l1, 5), "losses_finite": finite,
            "weights": "generator / VGG16 trunk: random init (no checkpoints offline); LPIPS heads: the reference's "
                       "lpips/weights/v0.1/vgg.pth"}


# ---------------------------------------------------------------------------------------------------
def plumbing_main(args, rank, world):
    """CPU stand-in used by tests/test_bench_launch.py: same launch / rendezvous / reduction structure as the
    GPU run on the gloo backend with a 8x8 generator; the JSON line says so (`plumbing: true`)."""
    import torch
    import torch.distributed as dist

    from stylerenderer_amd import distributed as sr_dist
    from stylerenderer_amd import model

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
    torch.manual_seed(0)
    torch.set_num_threads(2)
    g = model.Generator(8, 32, 2)
    sr_dist.freeze_unused_tail(g)
    net = sr_dist.construct_ddp(g, "cpu")
    gen = torch.Generator().manual_seed(1234 + rank)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        z = torch.randn(2, 32, generator=gen)
        for p_ in g.parameters():
            p_.grad = None
        img, _ = net([z])
        img.sum().backward()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    del net
    # the G+D iteration leg of the GPU run, same function: GraphedTrainer(capture=False), bucketed reducer, the
    # gradient_collective and all-reduce timing branches of N > 1
    train_res = None
    if not args.no_train:
        train_res = train_leg(torch.device("cpu"), rank, world, 2, 4, size=8, plumbing=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "generator fwd+bwd images/sec at 256^2", "plumbing": True, "n_gpus": world,
                          "value": round(2 * world * args.steps / elapsed, 2), "unit": "images/s",
                          "steps": args.steps, "warmup": 0, "data": "synthetic", "train_step": train_res}))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)" % (args.gpus, world))
    if args.plumbing:
        return plumbing_main(args, rank, world)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # a device tensor that would leave this library's kernels for a MIOpen / rocBLAS fallback raises (op._dispatch)
    os.environ.setdefault("SR_STRICT_NATIVE", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL kernels on high-priority streams: their own hardware queue, never behind the compute stream's kernels
        # (HIP multiplexes the streams of one priority over GPU_MAX_HW_QUEUES = 4 hardware queues).
        # Rendezvous limit 60 s: fewer than N ranks joining (a GPU that did not come up, a rank that died at import)
        # ends the job with a message instead of every rank sitting in the store until an outer limit.
        try:
            dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=60),
                                    pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
            seen = torch.ones(1, device=dev)
            dist.all_reduce(seen)                       # first RCCL collective: every rank's communicator is up
            ranks_seen = int(seen.item())
        except Exception as e:            # noqa: BLE001
            sys.stderr.write("bench.py: rank %d: the %d-rank RCCL group did not form within 60 s: %s\n" % (rank, world, e))
            if rank == 0:
                print(json.dumps({"metric": "generator fwd+bwd images/sec at 256^2", "value": None, "n_gpus": world,
                                  "rccl_ranks_seen": None, "error": "rendezvous of %d ranks failed within 60 s: %s" % (
                                      world, str(e)[:300])}), flush=True)
            os._exit(15)
        if ranks_seen != world:
            raise SystemExit("bench.py: %d ranks answered the first all-reduce, %d expected" % (ranks_seen, world))
        # later collectives (graph capture warm-ups, autotuning) keep the default generous limit

    from stylerenderer_amd import _lib, model
    from stylerenderer_amd import distributed as sr_dist
    from stylerenderer_amd.op import conv as conv_op

    _lib.lib()          # fail loudly if the HIP library is missing

    torch.manual_seed(0)
    g = model.Generator(args.size, 512, 8, channel_multiplier=2).to(dev)
    # the duplicated ToRGB tail never receives gradients: keep it out of the buckets
    sr_dist.freeze_unused_tail(g)
    net = sr_dist.construct_ddp(g, dev) if world > 1 else g
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    def step():
        z = torch.randn(args.batch, 512, device=dev, generator=gen)
        for p_ in g.parameters():
            p_.grad = None
        img, _ = net([z])
        img.sum().backward()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    conv_op.PROFILE = [] if rank == 0 else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof, conv_op.PROFILE = conv_op.PROFILE, None
    # ---- secondary, opt-in arithmetic (NOT the headline): the same step with the stride-2 convolutions and the stride-2 /
    # 1x1 weight gradients on the bf16 matrix cores, every fp32 operand split into three bf16 pieces (csrc/conv_wgrad_bf16x3.hip; error table in
    # profiles/r05_split_bf16.md: at or below the exact-fp32 MFMA kernels' under the same 2e-6 * sum|a||b| bar)
    split_res = None
    if world == 1 and not args.no_split_bf16:
        os.environ["SR_CONV_SPLIT_BF16"] = "1"
        try:
            for _ in range(2):
                step()
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            dt = time.perf_counter() - t1
            split_res = {"value": round(args.batch * args.steps / dt, 2), "unit": "images/s",
                         "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": "f32 operands split into 3 x bf16, six "
                         "bf16-MFMA products, fp32 accumulation (stride-2 3x3 convolution, its transposed form and the stride-2 / 1x1 weight gradients; the Winograd layers stay exact fp32)",
                         "opt_in": "SR_CONV_SPLIT_BF16=1", "speedup_vs_headline": round(elapsed / dt, 4)}
        finally:
            os.environ["SR_CONV_SPLIT_BF16"] = "0"
    per_rank_ms = [round(elapsed / args.steps * 1e3, 3)]
    if world > 1:
        mine = torch.zeros(world, device=dev, dtype=torch.float64)
        mine[rank] = elapsed
        dist.all_reduce(mine)                            # every rank's own clock (rank-indexed)
        per_rank_ms = [round(float(x) / args.steps * 1e3, 3) for x in mine.tolist()]
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    del net
    for p_ in g.parameters():
        p_.grad = None

    pmc_note = None
    if rank == 0 and world == 1 and not (args.no_pmc or args.pmc_child):
        # the timed region above is over: counters for the roofline objects below come from short --pmc passes over a
        # child of this script (never combined with trace domains; the parent's timings are not touched)
        pmc_note = collect_live_pmc()
    roof, breakdown, step_exec = None, None, None
    if rank == 0:
        # matrix-core flops the WHOLE step executes (every convolution / weight-gradient launch, Winograd launches
        # at 16/36 of their direct flops) over the step's wall time — the time-weighted companion of roofline.frac
        # (which is the dominant kernel alone) and of model_flop_frac_of_mfma_peak (direct-convolution flops)
        exe_step = sum(executed_flops(kind, geom, fl) for (kind, geom, fl, _e0, _e1) in prof) / max(args.steps, 1)
        step_exec = {"executed_gflop_per_step": round(exe_step / 1e9, 1),
                     "executed_tflops": round(exe_step / (elapsed / args.steps) / 1e12, 2),
                     "frac": round(exe_step / (elapsed / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
        # ---- roofline of the dominant kernel: the stride-1 3x3 convolution (forward and data-gradient
        # launches of the 64^2..256^2 layers; k_conv_wino, or k_conv_mfma with SR_WINOGRAD=0), timed
        # inside the steps above
        dom = [(fl, e0.elapsed_time(e1)) for (kind, geom, fl, e0, e1) in prof
               if kind == "conv" and geom[0] == 3 and geom[1] == 1 and geom[2] == 0 and geom[7] > 16]
        by_kind = {}
        for (kind, geom, fl, e0, e1) in prof:
            k = "%s_k%d_s%d_t%d" % (kind, geom[0], geom[1], geom[2])
            a = by_kind.setdefault(k, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1)
            a[2] += 1
        if dom:
            fl = sum(d[0] for d in dom) / len(dom)
            ms = sum(d[1] for d in dom) / len(dom)
            alg = fl / (ms * 1e-3) / 1e12
            wino = os.environ.get("SR_WINOGRAD", "1") != "0"
            # Winograd F(2x2,3x3) issues 16/36 of the direct-convolution multiplies on the matrix cores:
            # `achieved` / `frac` are the EXECUTED MFMA flops (what the 157.3 TFLOP/s roof bounds); the
            # algorithmic (direct-convolution) rate is reported beside it
            exe = alg * 16.0 / 36.0 if wino else alg
            roof = {"bound": "mfma", "kernel": "k_conv_wino<8>" if wino else "k_conv_mfma<1,3,3,32,4,1>",
                    "achieved": round(exe, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(exe / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None, "launches": len(dom),
                    "avg_launch_ms": round(ms, 4), "flop_per_launch": fl,
                    "algorithmic_tflops": round(alg, 2),
                    "algorithmic_frac": round(alg / FP32_MFMA_PEAK_TFLOPS, 4)}
            if wino:
                roof["note"] = ("Winograd F(2x2,3x3): achieved/frac = MFMA flops actually issued (algorithmic x16/36); "
                                "algorithmic_* = direct-convolution flops of SURVEY 8(d) / time (can exceed the roof)")
            roof.update(pmc_traffic("k_conv_wino<8>" if wino else "k_conv_mfma<1; 3; 3; 32; 4; 1; true>"))
        breakdown = {k: {"ms_per_step": round(v[1] / args.steps, 3),
                         "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1) if v[1] > 0 else None,
                         "launches_per_step": v[2] // max(args.steps, 1)} for k, v in sorted(by_kind.items())}
    del g
    torch.cuda.empty_cache()

    def leg(fn, *a, **k):
        """A secondary leg that raises is REPORTED in the line ({"error": ...}), it does not take the headline
        measurement above down with it.  At N > 1 a leg that fails on ONE rank (an out-of-memory, a timed-out signal)
        would leave the others blocked in collectives: that rank aborts the whole job at once (_fail_fast; rank 0
        still prints the headline it holds, also when a peer's exit makes the launcher terminate it)."""
        try:
            return fn(*a, **k)
        except Exception as e:           # noqa: BLE001
            import traceback

            text = traceback.format_exc()
            if world > 1:
                _fail_fast(rank, world, {"train_leg": "train_step", "raster_leg": "rasterizer",
                                         "inversion_leg": "inversion"}.get(fn.__name__, fn.__name__), text)
            sys.stderr.write(text)
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}

    if world > 1 and rank == 0:
        import signal

        _STASH["line"] = {"metric": "generator fwd+bwd images/sec at 256^2",
                          "value": round(args.batch * world * args.steps / elapsed, 2), "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "BASELINE config[1]: 256x256 Generator fwd+bwd, batch %d per GPU, "
                                                 "random latents" % args.batch, "global_batch": args.batch * world,
                                     "size": args.size, "parallelism": "dp%d" % world},
                          "rccl_ranks_seen": ranks_seen, "ms_per_step_per_rank": per_rank_ms,
                          "roofline": roof, "kernel_breakdown": breakdown}
        signal.signal(signal.SIGTERM, _on_sigterm)

    train_res = None
    if not args.no_train:
        train_res = leg(train_leg, dev, rank, world, args.train_iters, args.train_batch, args.size)
    raster = None
    if not args.no_raster and rank == 0:
        raster = leg(raster_leg, dev, 1, cpu_baseline=(world == 1 and not args.no_cpu_baseline))
    inversion_res = None
    if not args.no_inversion and rank == 0 and world == 1:
        inversion_res = leg(inversion_leg, dev, args.inversion_steps, args.size)
    if rank == 0 and world == 1 and not args.pmc_child:
        # whole-step HBM traffic of the two replayed workloads: collected live with --step-traffic (separate --pmc passes
        # over their probes in scripts/, minutes of counter collection), otherwise read from the committed summary
        CAD = {"d": 1.0, "r1": 1.0 / 16, "g": 1.0, "path": 1.0 / 4, "d_opt": 1.0 + 1.0 / 16, "g_opt": 1.0 + 1.0 / 4, "ema": 1.0}
        ORDER = ("d", "r1", "g", "path", "d_opt", "g_opt", "ema")
        live = {}
        if args.step_traffic and not args.no_pmc:
            if isinstance(train_res, dict) and train_res.get("phases"):
                ph = train_res["phases"]
                per, note = graph_step_traffic("train_phase_pmc_probe.py", [args.train_batch],
                                               [(n, int(ph[n].get("kernel_nodes") or 0)) for n in ORDER])
                live["train_step"] = ({"phases": {k: round(v) for k, v in per.items()},
                                       "bytes_per_iteration": round(sum(per[n] * CAD[n] for n in ORDER))} if per else
                                      {"error": note})
            if isinstance(inversion_res, dict) and inversion_res.get("roofline"):
                nodes = int(inversion_res["roofline"].get("kernel_launches_per_step") or 0)
                per, note = graph_step_traffic("inversion_replay_probe.py", [10], [("steps", 10 * nodes)])
                live["inversion"] = {"bytes_per_step": round(per["steps"] / 10.0)} if per else {"error": note}
            live["source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes spawned by `bench.py --step-traffic` over "
                              "scripts/train_phase_pmc_probe.py / scripts/inversion_replay_probe.py (all kernels of the "
                              "replays, 2*FETCH_SIZE+WRITE_SIZE, KB)")
            if args.step_traffic != "1":
                with open(args.step_traffic, "w") as f:
                    json.dump(live, f, indent=1)
        table, fname = (live, None) if live else committed_step_traffic()
        if table:
            src = table.get("source", "") if fname is None else "profiles/%s (%s)" % (fname, table.get("source", ""))
            t = (table.get("train_step") or {}).get("bytes_per_iteration")
            if t and isinstance(train_res, dict) and train_res.get("roofline"):
                r = train_res["roofline"]
                r.update({"traffic": t, "traffic_unit": "bytes/iteration (every kernel of each captured phase x its cadence)",
                          "traffic_measured_in_this_run": fname is None, "traffic_source": src,
                          "traffic_by_phase": (table.get("train_step") or {}).get("phases"),
                          "hbm_GBps_at_measured_iteration": round(t / (train_res["ms_per_iter"] * 1e-3) / 1e9, 1)})
            t = (table.get("inversion") or {}).get("bytes_per_step")
            if t and isinstance(inversion_res, dict) and inversion_res.get("roofline"):
                r = inversion_res["roofline"]
                r.update({"traffic": t, "traffic_unit": "bytes/step (every kernel of the replayed step)",
                          "traffic_measured_in_this_run": fname is None, "traffic_source": src})
                if r.get("ms_per_replay"):
                    r["hbm_GBps_at_measured_step"] = round(t / (r["ms_per_replay"] * 1e-3) / 1e9, 1)
    result = None
    if rank == 0:
        images = args.batch * world * args.steps
        value = images / elapsed
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import model_oracle

            # grouped convolutions scale poorly past a few dozen threads (256 threads: 50x slower than 32 on the GPU
            # box's host): a short sweep (1 warm-up + 1 timed iteration per thread count) picks the host's best
            # configuration, the full sample then runs there; the whole leg stays within ~30-40 s
            ncpu = os.cpu_count() or 1
            sweep = {}
            for th in sorted({min(ncpu, int(x)) for x in str(args.cpu_threads).split(",")}):
                torch.set_num_threads(th)
                sweep[th] = round(model_oracle.time_generator_fwd_bwd(args.size, args.cpu_batch, 1, 1), 4)
                if sweep[th] < 0.5 * max(sweep.values()):
                    break                               # past the knee: larger counts only get slower
            best = max(sweep, key=sweep.get)
            torch.set_num_threads(best)
            v = model_oracle.time_generator_fwd_bwd(args.size, args.cpu_batch, args.cpu_iters, 1)
            cpu = {"value": round(v, 4), "unit": "images/s", "cores": torch.get_num_threads(),
                   "kind": "port", "thread_sweep_images_per_s": {str(k): x for k, x in sweep.items()},
                   "sample": "oracle/model_oracle.py Generator(%d) fwd+bwd, batch %d, 1 warm-up + %d timed "
                             "iteration(s) at the best of a thread sweep %s (1 + 1 iterations each): %d torch threads "
                             "of %d host cores"
                             % (args.size, args.cpu_batch, args.cpu_iters, sorted(sweep), torch.get_num_threads(),
                                ncpu)}
        result = {
            "metric": "generator fwd+bwd images/sec at 256^2", "value": round(value, 2),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config[1]: 256x256 Generator fwd+bwd, batch %d per GPU, random "
                                   "latents" % args.batch,
                       "global_batch": args.batch * world, "size": args.size,
                       "parallelism": "dp%d" % world,
                       "step": "zero_grad + forward + backward" + (" + DDP all-reduce (RCCL)" if world > 1 else "")},
            "rccl_ranks_seen": ranks_seen, "ms_per_step_per_rank": per_rank_ms,
            "model_flop_frac_of_mfma_peak": round(value / world * FLOP_PER_IMAGE_FWD_BWD / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4),
            "step_executed_mfma_frac": step_exec["frac"] if step_exec else None, "step_executed_mfma": step_exec,
            "roofline": roof, "cpu_baseline": cpu, "kernel_breakdown": breakdown, "pmc_note": pmc_note,
            "split_bf16_variant": split_res,
            "train_step": train_res, "rasterizer": raster, "inversion": inversion_res,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
