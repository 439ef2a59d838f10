"""Per-workgroup phase timing of k_convt_fused (debug build: scripts/build_variant.sh ct conv_mfma.hip -DCONVT_TIMING).
Run:  STYLERENDERER_AMD_LIB=build/mb/libsr_ct.so python scripts/convt_timing.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylerenderer_amd.op.conv import conv2d_mfma

cdll = ctypes.CDLL(os.environ["STYLERENDERER_AMD_LIB"])
dev = "cuda"
for (b, c, n, res) in ((16, 256, 128, 128), (16, 512, 256, 64), (16, 512, 512, 32)):
    x = torch.randn(b, c, res, res, device=dev); wt = torch.randn(9, c, n, device=dev)
    isc = torch.randn(b, c, device=dev); osc = torch.randn(b, n, device=dev)
    for _ in range(3):
        y = conv2d_mfma(x, wt, isc, osc, None, 3, 2, 0, True)
    torch.cuda.synchronize()
    m = 16384
    buf = np.zeros(m * 8, dtype=np.int64)
    cdll.sr_debug_convt_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(m * 8))
    s = buf.reshape(m, 8)
    nwg = b * (res // 32) * (res // 4) * ((n + 127) // 128)
    s = s[:min(nwg, m)]
    t = s[:, :4].astype(np.float64) * 10.0
    ph = np.diff(t, axis=1)
    mhz = s[:, 5].astype(np.float64) / ((t[:, 3] - t[:, 0]) / 1e3)
    span = (t[:, 3].max() - t[:, 0].min()) / 1e3
    chunks = c // 8
    print("B%d C%d N%d res%d: %d workgroups; prologue %.2f us, loop %.2f us (%.3f us = %.0f cycles per chunk of 144 MFMAs = 9216), epilogue %.2f us; clock %.0f MHz; span %.1f us, sum of workgroup times / 256 = %.1f us" % (
        b, c, n, res, len(s), np.median(ph[:, 0]) / 1e3, np.median(ph[:, 1]) / 1e3, np.median(ph[:, 1]) / 1e3 / chunks,
        np.median(ph[:, 1]) / 1e3 / chunks * np.median(mhz), np.median(ph[:, 2]) / 1e3, np.median(mhz), span, (t[:, 3] - t[:, 0]).sum() / 1e3 / 256))
