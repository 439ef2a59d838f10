#!/usr/bin/env python3
"""Benchmark of the StyleRenderer generator hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Metric (BASELINE.json): generator fwd+bwd images/s at 256x256.  Workload = BASELINE config[1]:
`Generator(256, 512, 8, channel_multiplier=2)` forward + backward, batch 16 per GPU, fresh random
latents every step, fresh per-layer noise, loss = image.sum(), fp32 end to end (exact-fp32 MFMA).
Inputs are generated on the device (resident in HBM when the timed region starts).
One process per GPU; N > 1 is data parallel (weak scaling: 16 images per GPU) with the gradient
all-reduce over RCCL overlapped with backward (torch DDP, nccl backend == RCCL on ROCm).

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  roofline      achieved TFLOP/s of the dominant kernel (3x3 stride-1 MFMA convolution, forward and
                data-gradient launches) measured with events on the launch stream INSIDE the timed
                steps, against the 157.3 TFLOP/s fp32-MFMA peak of gfx950;
  cpu_baseline  the CPU oracle (oracle/model_oracle.py, a restatement of the reference's grouped-
                convolution formulation) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
FLOP_PER_IMAGE_FWD_BWD = 270.7e9       # 3 x 90.24 GFLOP of convolution per image


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-raster", action="store_true", help="skip the rasterizer (BASELINE config[3]) leg")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-iters", type=int, default=6)
    ap.add_argument("--cpu-threads", type=int, default=32)
    return ap.parse_args()


def raster_leg(dev, batch=64, res=256, iters=5):
    """Second half of the metric: rasterizer Mtri/s on BASELINE config[3] (BFM-size-class mesh,
    ~50k triangles, 256x256, batch 64, int64 ids as the API demands), forward and forward+backward
    (fused attribute interpolation + fused gradient scatter).  Algorithmic HBM bytes per image
    24*nf + 12*nv + 36*h*w."""
    import stylerenderer_amd.op as op
    from stylerenderer_amd import synth

    v0, tri = synth.face_sized_mesh()
    v = torch.from_numpy(synth.random_poses(v0, batch, seed=1234)).to(dev)
    nrm = torch.from_numpy(synth.vertex_normals(v.cpu().numpy(), tri)).to(dev)
    t = torch.from_numpy(tri).to(dev)
    nf, nv = tri.shape[0], v0.shape[0]

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    ms_f = timed(lambda: op.rasterize(v, nrm, t, res))
    vg, ng = v.clone().requires_grad_(), nrm.clone().requires_grad_()

    def fb():
        vg.grad = ng.grad = None
        op.rasterize(vg, ng, t, res).sum().backward()

    ms_fb = timed(fb)
    bytes_img = 24 * nf + 12 * nv + 36 * res * res
    return {"workload": "BASELINE config[3]: nv=%d nf=%d, %dx%d, batch %d" % (nv, nf, res, res, batch),
            "fwd_mtri_s": round(batch * nf / ms_f / 1e3, 1), "fwd_ms": round(ms_f, 3),
            "fwd_bwd_mtri_s": round(batch * nf / ms_fb / 1e3, 1), "fwd_bwd_ms": round(ms_fb, 3),
            "fwd_algorithmic_GBps": round(batch * bytes_img / ms_f / 1e6, 1),
            "bit_exact_vs_cpu_oracle": "tests/test_ops_gpu.py"}


def pmc_traffic(kernel_row):
    """HBM bytes per launch of `kernel_row` from the newest committed PMC summary (profiles/r*_pmc.csv,
    written by scripts/profile_round.sh + scripts/pmc_summary.py from separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this same command).  Units are KB; gfx950's FETCH_SIZE reports
    half of a 16 B/lane streaming read (MI355X_MICROARCH.md, HBM section; confirmed here on k_nba_bwd
    and k_rowdot, whose read:write byte ratios are known), hence 2*FETCH + WRITE."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.csv")))
    if not files:
        return {}
    with open(files[-1]) as f:
        for r in csv.DictReader(l for l in f if not l.startswith("#")):
            if r["kernel"] == kernel_row and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
                b = (2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0
                return {"traffic": round(b), "traffic_unit": "bytes/launch",
                        "traffic_source": "profiles/%s (2*FETCH_SIZE+WRITE_SIZE)" % os.path.basename(files[-1])}
    return {}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from stylerenderer_amd import _lib, model
    from stylerenderer_amd.op import conv as conv_op

    _lib.lib()          # fail loudly if the HIP library is missing

    torch.manual_seed(0)
    g = model.Generator(args.size, 512, 8, channel_multiplier=2).to(dev)
    # the duplicated ToRGB tail never receives gradients: keep it out of DDP buckets
    used = len(g.to_rgbs) // 2
    for m in list(g.to_rgbs)[used:]:
        for p_ in m.parameters():
            p_.requires_grad_(False)
    net = g
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(
            g, device_ids=[local_rank], broadcast_buffers=False, bucket_cap_mb=32,
            gradient_as_bucket_view=True)
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
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        images = args.batch * world * args.steps
        value = images / elapsed
        # ---- roofline of the dominant kernel: the stride-1 3x3 convolution (forward and data-gradient
        # launches of the 64^2..256^2 layers; k_conv_wino, or k_conv_mfma with SR_WINOGRAD=0), timed
        # inside the steps above
        dom = [(fl, e0.elapsed_time(e1)) for (kind, geom, fl, e0, e1) in prof
               if kind == "conv" and geom[0] == 3 and geom[1] == 1 and geom[2] == 0 and geom[7] > 16]
        roof = None
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
            ach = fl / (ms * 1e-3) / 1e12
            wino = os.environ.get("SR_WINOGRAD", "1") != "0"
            roof = {"bound": "mfma", "kernel": "k_conv_wino<8>" if wino else "k_conv_mfma<1,3,3,32,4,1>",
                    "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None, "launches": len(dom),
                    "avg_launch_ms": round(ms, 4), "flop_per_launch": fl}
            if wino:
                # `achieved` counts the ALGORITHMIC (direct-convolution) FLOPs of SURVEY.md 8(d); the
                # Winograd F(2x2,3x3) kernel issues 16/36 of them on the matrix cores
                roof["executed_tflops"] = round(ach * 16.0 / 36.0, 2)
                roof["executed_frac"] = round(ach * 16.0 / 36.0 / FP32_MFMA_PEAK_TFLOPS, 4)
                roof["note"] = ("Winograd F(2x2,3x3): achieved = direct-conv FLOPs / time, so frac can exceed 1; "
                                "executed_* = MFMA FLOPs actually issued (x16/36)")
            roof.update(pmc_traffic("k_conv_wino<8>" if wino else "k_conv_mfma<1; 3; 3; 32; 4; 1; true>"))
        breakdown = {k: {"ms_per_step": round(v[1] / args.steps, 3),
                         "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1) if v[1] > 0 else None,
                         "launches_per_step": v[2] // max(args.steps, 1)} for k, v in sorted(by_kind.items())}
        raster = raster_leg(dev) if not args.no_raster else None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import model_oracle

            # grouped convolutions scale poorly past a few dozen threads (256 threads: 50x slower
            # than 32 on the GPU box's host); the sample stays within ~30 s
            torch.set_num_threads(min(os.cpu_count() or 1, args.cpu_threads))
            v = model_oracle.time_generator_fwd_bwd(args.size, args.cpu_batch, args.cpu_iters, 1)
            cpu = {"value": round(v, 4), "unit": "images/s", "cores": torch.get_num_threads(),
                   "kind": "port",
                   "sample": "oracle/model_oracle.py Generator(%d) fwd+bwd, batch %d, 1 warm-up + %d timed "
                             "iteration(s), %d torch threads of %d host cores"
                             % (args.size, args.cpu_batch, args.cpu_iters, torch.get_num_threads(),
                                os.cpu_count() or 1)}
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
            "model_flop_frac_of_mfma_peak": round(value / world * FLOP_PER_IMAGE_FWD_BWD / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4),
            "roofline": roof, "cpu_baseline": cpu, "kernel_breakdown": breakdown, "rasterizer": raster,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
