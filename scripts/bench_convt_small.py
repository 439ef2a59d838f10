"""Transposed convolution at small batch / small maps: fused four-phase kernel (SR_CONVT_FUSED=1 forces it) vs per-phase
split-K launches (=0), timed inside a captured graph (what the training / inversion loops replay)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from stylerenderer_amd import graphs
    from stylerenderer_amd.op.conv import conv2d_mfma

    tag = {"1": "fused ", "0": "phases"}.get(os.environ.get("SR_CONVT_FUSED", ""), "auto  ")
    tag += {"0": " taps off", "1": " taps forced"}.get(os.environ.get("SR_CONVT_TAPS", ""), " taps auto")
    tag += {"0": " (patch form)"}.get(os.environ.get("SR_CONVT_TAPS_GEMM", ""), " (flattened)")
    for (b, c, n, res) in ((1, 512, 512, 32), (2, 512, 512, 32), (4, 512, 512, 32), (8, 512, 512, 32), (1, 512, 256, 64),
                           (2, 512, 256, 64), (4, 512, 256, 64), (1, 256, 128, 128), (2, 256, 128, 128), (4, 256, 128, 128),
                           (4, 512, 512, 16), (8, 512, 512, 16), (4, 512, 512, 8), (4, 512, 512, 4)):
        x = torch.randn(b, c, res, res, device="cuda")
        wt = torch.randn(9, c, n, device="cuda")
        isc = torch.randn(b, c, device="cuda")
        osc = torch.randn(b, n, device="cuda")

        def body():
            for _ in range(10):
                conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, True)

        body()
        torch.cuda.synchronize()
        g = graphs.capture(body)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / 30 * 1e-3
        print("%s convT B%d C%d N%d res%d: %.3f ms  %.1f TFLOP/s  (%d nodes)" % (
            tag, b, c, n, res, dt * 1e3, 2.0 * b * res * res * c * n * 9 / dt / 1e12, g.kernel_nodes // 10), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        # fused four-phase kernel forced; per-phase launches; the default choice (tap-split for small maps)
        for env in ({"SR_CONVT_FUSED": "1", "SR_CONVT_TAPS": "0"}, {"SR_CONVT_FUSED": "0", "SR_CONVT_TAPS": "0"},
                    {"SR_CONVT_FUSED": "0", "SR_CONVT_TAPS": "1", "SR_CONVT_TAPS_GEMM": "0"},
                    {"SR_CONVT_FUSED": "0", "SR_CONVT_TAPS": "1"}, {}):
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env),
                           check=False, timeout=600)
