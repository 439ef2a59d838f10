#!/usr/bin/env python3
"""GPU box: the split-bf16 spike (csrc/conv_wgrad_bf16x3.hip, SR_CONV_SPLIT_BF16=1) against the exact-fp32 MFMA weight
gradient of the 1x1 convolution on the discriminator's skip-convolution shapes: milliseconds, TFLOP/s-equivalent
(direct-convolution flops / time) and the error of both against float64, as a histogram of |error| / sum |a||b|.
Writes markdown to stdout (profiles/r05_split_bf16.md)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from stylerenderer_amd.op.conv import conv2d_wgrad_mfma  # noqa: E402

DEV = "cuda"


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "s2only":
        return stride2(reps=3, shapes=((16, 256, 128, 128, True), (16, 512, 256, 64, True)))
    if len(sys.argv) > 1 and sys.argv[1] == "tonly":
        return conv_t(reps=5)
    if len(sys.argv) > 1 and sys.argv[1] == "convonly":
        return conv_s2(reps=3, shapes=((16, 128, 256, 257), (16, 256, 512, 129)))
    g = torch.Generator().manual_seed(1)
    rows, hist = [], {}
    edges = [0, 1e-8, 3e-8, 1e-7, 3e-7, 1e-6, 2e-6, 1e-5, 1.0]
    for b, c, n, res in ((16, 128, 256, 128), (16, 256, 512, 64), (16, 512, 512, 32), (8, 128, 256, 128), (8, 256, 512, 64),
                         (16, 3, 128, 256), (16, 512, 512, 64)):
        x = torch.randn(b, c, res, res, generator=g).to(DEV)
        gy = torch.randn(b, n, res, res, generator=g).to(DEV)
        xs, gs = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)
        flops = 2.0 * b * res * res * c * n
        out = {}
        for mode in ("0", "1"):
            os.environ["SR_CONV_SPLIT_BF16"] = mode
            out[mode] = conv2d_wgrad_mfma(x, gy, xs, gs, 1, 1, 0)
            ms = timed(lambda: conv2d_wgrad_mfma(x, gy, xs, gs, 1, 1, 0))
            out[mode + "ms"] = ms
        # float64 reference on a channel sub-block (the full product is 1e12 flops in fp64)
        cu, cv = min(c, 64), min(n, 64)
        xd = (x[:, :cu].double() * xs[:, :cu].double()[:, :, None, None]).flatten(2)
        gd = (gy[:, :cv].double() * gs[:, :cv].double()[:, :, None, None]).flatten(2)
        want = torch.einsum("bcp,bnp->cn", xd, gd)
        mag = torch.einsum("bcp,bnp->cn", xd.abs(), gd.abs())
        errs = {}
        for mode in ("0", "1"):
            e = ((out[mode][0, :cu, :cv].double() - want).abs() / mag).flatten().cpu().numpy()
            errs[mode] = e
            h = hist.setdefault(mode, np.zeros(len(edges) - 1, np.int64))
            h += np.histogram(e, bins=edges)[0]
        rows.append((b, c, n, res, out["0ms"], flops / out["0ms"] / 1e9, out["1ms"], flops / out["1ms"] / 1e9,
                     out["0ms"] / out["1ms"], errs["0"].max(), np.sqrt((errs["0"] ** 2).mean()), errs["1"].max(),
                     np.sqrt((errs["1"] ** 2).mean())))
    print("# Split-bf16 spike: 1x1 weight gradient, exact-fp32 MFMA (`k_wgrad_mfma<1,1,1,...>`) vs three-way bf16 split "
          "(`k_wgrad1_bf16x3`), both incl. their `k_wgrad_reduce`\n")
    print("| B | Cin | Cout | map | fp32 MFMA ms | TFLOP/s | split-bf16 ms | TFLOP/s-equivalent | speed-up | fp32 max err | "
          "fp32 rms | split max err | split rms |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %d | %d | %d^2 | %.4f | %.1f | %.4f | %.1f | **%.2fx** | %.2e | %.2e | %.2e | %.2e |" % r)
    print("\nErrors are |result - float64| / sum |a||b| over a 64 x 64 channel block of every shape (the bar of "
          "tests/test_conv_gpu.py is 2e-6).\n")
    print("| |error| / sum|a||b| | " + " | ".join("< %g" % e for e in edges[1:]) + " |")
    print("|---|" + "---|" * (len(edges) - 1))
    for mode, name in (("0", "exact-fp32 MFMA"), ("1", "split-bf16 (6 products)")):
        print("| %s | " % name + " | ".join(str(int(v)) for v in hist[mode]) + " |")
    stride2()
    conv_s2()
    conv_t()


def conv_t(reps=20, shapes=None):
    """The stride-2 transposed 3x3 convolution (k_convt_fused + strips vs k_split_w_t + k_convt_bf16x3 + strips)."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    g = torch.Generator().manual_seed(4)
    rows = []
    for b, c, n, res in shapes or ((16, 256, 128, 128), (16, 512, 256, 64), (16, 512, 512, 32), (4, 256, 128, 128),
                                   (4, 512, 256, 64)):
        x = torch.randn(b, c, res, res, generator=g).to(DEV)
        wt = (torch.randn(9, c, n, generator=g) / (3 * c ** 0.5)).to(DEV)
        isc, osc = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)
        flops = 2.0 * b * res * res * c * n * 9
        ms, res_t = {}, {}
        for mode in ("0", "1"):
            os.environ["SR_CONV_SPLIT_BF16"] = mode
            res_t[mode] = conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, True)
            ms[mode] = timed(lambda: conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, True), reps)
        diff = float((res_t["1"] - res_t["0"]).abs().max() / res_t["0"].abs().max())
        rows.append((b, c, n, res, ms["0"], flops / ms["0"] / 1e9, ms["1"], flops / ms["1"] / 1e9, ms["0"] / ms["1"], diff))
    print("\n# Stride-2 transposed 3x3 convolution: `k_convt_fused` (exact fp32 MFMA) vs `k_split_w_t` + `k_convt_bf16x3`, "
          "both incl. the fp32 border strips\n")
    print("| B | Cin | Cout | input | fp32 MFMA ms | TFLOP/s | split-bf16 ms | TFLOP/s-equivalent | speed-up | "
          "max |split - fp32| / max |fp32| |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %d | %d | %d^2 | %.4f | %.1f | %.4f | %.1f | **%.2fx** | %.1e |" % r)


def conv_s2(reps=20, shapes=None):
    """The stride-2 3x3 convolution itself (k_conv_mfma<2,3,3,...> vs k_split_w_s2 + k_conv_s2_bf16x3)."""
    from stylerenderer_amd.op.conv import conv2d_mfma

    g = torch.Generator().manual_seed(3)
    rows = []
    # (B, Cin, Cout, input): data gradient of the headline's up-sampling layers, the discriminator's down-sampling ones
    for b, c, n, res in shapes or ((16, 128, 256, 257), (16, 256, 512, 129), (16, 512, 512, 65), (4, 128, 256, 257),
                                   (8, 128, 256, 257), (8, 256, 512, 129)):
        out = (res - 3) // 2 + 1
        x = torch.randn(b, c, res, res, generator=g).to(DEV)
        wt = (torch.randn(9, c, n, generator=g) / (3 * c ** 0.5)).to(DEV)
        isc, osc = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)
        flops = 2.0 * b * out * out * c * n * 9
        ms, res_t = {}, {}
        for mode in ("0", "1"):
            os.environ["SR_CONV_SPLIT_BF16"] = mode
            res_t[mode] = conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, False)
            ms[mode] = timed(lambda: conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, False), reps)
        diff = float((res_t["1"] - res_t["0"]).abs().max() / res_t["0"].abs().max())
        rows.append((b, c, n, res, ms["0"], flops / ms["0"] / 1e9, ms["1"], flops / ms["1"] / 1e9, ms["0"] / ms["1"], diff))
    print("\n# Stride-2 3x3 convolution: `k_conv_mfma<2,3,3,...>` (exact fp32 MFMA) vs `k_split_w_s2` + `k_conv_s2_bf16x3`\n")
    print("| B | Cin | Cout | input | fp32 MFMA ms | TFLOP/s | split-bf16 ms | TFLOP/s-equivalent | speed-up | "
          "max |split - fp32| / max |fp32| |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %d | %d | %d^2 | %.4f | %.1f | %.4f | %.1f | **%.2fx** | %.1e |" % r)


def stride2(reps=20, shapes=None):
    """The stride-2 3x3 weight gradient of the up- / down-sampling layers (k_wgrad_s2_dma vs k_wgrad_s2_bf16x3)."""
    g = torch.Generator().manual_seed(2)
    rows = []
    # (B, C, N, res, transposed): headline up-sampling layers, the training step's batch, the discriminator's c3s2
    for b, c, n, res, tr in shapes or ((16, 256, 128, 128, True), (16, 512, 256, 64, True), (16, 512, 512, 32, True),
                                       (4, 256, 128, 128, True), (4, 512, 256, 64, True), (8, 128, 256, 257, False),
                                       (8, 256, 512, 129, False)):
        out = 2 * res + 1 if tr else (res - 3) // 2 + 1
        x = torch.randn(b, c, res, res, generator=g).to(DEV)
        gy = torch.randn(b, n, out, out, generator=g).to(DEV)
        xs, gs = torch.randn(b, c, generator=g).to(DEV), torch.randn(b, n, generator=g).to(DEV)
        grid = res if tr else out
        flops = 2.0 * b * grid * grid * c * n * 9
        ms, res_t = {}, {}
        for mode in ("0", "1"):
            os.environ["SR_CONV_SPLIT_BF16"] = mode
            res_t[mode] = conv2d_wgrad_mfma(x, gy, xs, gs, 3, 2, 0, tr)
            ms[mode] = timed(lambda: conv2d_wgrad_mfma(x, gy, xs, gs, 3, 2, 0, tr), reps)
        diff = float((res_t["1"] - res_t["0"]).abs().max() / res_t["0"].abs().max())
        rows.append((b, c, n, res, "transposed" if tr else "strided", ms["0"], flops / ms["0"] / 1e9, ms["1"],
                     flops / ms["1"] / 1e9, ms["0"] / ms["1"], diff))
    print("\n# Stride-2 3x3 weight gradient: `k_wgrad_s2_dma` (exact fp32 MFMA) vs `k_wgrad_s2_bf16x3`, both incl. "
          "`k_wgrad_reduce`\n")
    print("| B | Cin | Cout | input | kind | fp32 MFMA ms | TFLOP/s | split-bf16 ms | TFLOP/s-equivalent | speed-up | "
          "max |split - fp32| / max |fp32| |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %d | %d | %d | %d^2 | %s | %.4f | %.1f | %.4f | %.1f | **%.2fx** | %.1e |" % r)


if __name__ == "__main__":
    main()
