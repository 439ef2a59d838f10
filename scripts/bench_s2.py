import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from stylerenderer_amd.op.conv import conv2d_mfma
dev = 'cuda'
def run(b, c, n, res, iters=10):
    x = torch.randn(b, c, res, res, device=dev); wt = torch.randn(9, c, n, device=dev)
    isc = torch.randn(b, c, device=dev); osc = torch.randn(b, n, device=dev)
    for _ in range(3): conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, False)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(iters): conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, False)
    torch.cuda.synchronize(); dt = (time.time() - t) / iters
    oh = (res - 3) // 2 + 1
    fl = 2 * b * oh * oh * c * n * 9
    print("ROT=%s B%d C%d N%d res%d: %.3f ms %.1f TFLOP/s" % (os.environ.get("SR_CONV_ROT", "1"), b, c, n, res, dt * 1e3, fl / dt / 1e12), flush=True)
run(16, 128, 256, 257); run(16, 256, 512, 129); run(16, 512, 512, 65); run(4, 128, 256, 257)
