import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from stylerenderer_amd import graph_train, train
dev = torch.device("cuda", 0)
faces = train.SyntheticFaceSource(dev, seed=0)
tr = graph_train.GraphedTrainer(size=256, latent=512, n_mlp=8, use_mesh=True, device=dev, seed=0, batch=4, mesh_vertices=faces.model.dim[2] // 3)
data = train.SyntheticImages(64, 256, dev)
for _ in range(3): tr.step(data.batch(4), faces=faces, log=False)
torch.cuda.synchronize()
for k in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(data.batch(4), faces=faces, log=False)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("iteration %d: host %.2f ms, until idle %.2f ms" % (tr.iteration, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
# pieces
torch.cuda.synchronize(); t0 = time.perf_counter(); b = data.batch(4); t1 = time.perf_counter(); torch.cuda.synchronize()
print("data.batch host %.2f ms" % ((t1 - t0) * 1e3))
for name, g in tr.graphs.items():
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("graph %s: replay() host %.2f ms, done %.2f ms" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
