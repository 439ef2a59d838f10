"""Child process of tests/test_graph_trainer_gpu.py::test_overlap_modes_under_constrained_hardware_queues.

Runs three iterations of a small GraphedTrainer (64^2, mesh maps, 4 images) on a ONE-rank RCCL group with the bucketed
gradient reduction forced on, under whatever GPU_MAX_HW_QUEUES / SR_GRAD_OVERLAP the parent put in the environment
(HIP reads the queue count when the runtime starts, hence a process of its own).  Prints one JSON line: the release mode
the reducer chose, its probe, a checksum of both parameter buffers, and the overlap measurement of the D phase."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    from stylerenderer_amd import graph_train, train

    port = int(sys.argv[1])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                            pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
    dev = torch.device("cuda")
    faces = train.SyntheticFaceSource(dev, seed=0, face_sized=False)
    tr = graph_train.GraphedTrainer(size=64, latent=64, n_mlp=2, channel_multiplier=2, use_mesh=True, device=dev, seed=0,
                                    batch=4, mesh_vertices=faces.model.dim[2] // 3, force_collectives=True)
    data = train.SyntheticImages(8, 64, dev)
    torch.manual_seed(123)
    tr.np_rng = np.random.RandomState(5)
    batches = [data.batch(4) for _ in range(3)]
    meshes = [tuple(t.clone() for t in faces.sample(4)) for _ in range(3)]
    torch.manual_seed(321)
    logs = [tr.step(x, mesh=m) for x, m in zip(batches, meshes)]
    ov = tr.measure_overlap("d")
    torch.cuda.synchronize()
    tr.reduce_d.check()
    tr.reduce_g.check()
    digest = hashlib.sha256(tr.g_optim.flat_p.cpu().numpy().tobytes() + tr.d_optim.flat_p.cpu().numpy().tobytes())
    print("CHILD " + json.dumps({
        "release": tr.reduce_d.release, "probe": tr.reduce_d.release_probe, "digest": digest.hexdigest(),
        "finite": bool(all(np.isfinite(v) for log in logs for v in log.values())), "overlap": ov,
        "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "forced": os.environ.get("SR_GRAD_OVERLAP")}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
