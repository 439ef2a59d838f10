"""Per-workgroup phase timing of k_wgrad_wino (debug build: scripts/build_variant.sh t conv_wgrad_wino.hip -DWGW_TIMING).
Run:  STYLERENDERER_AMD_LIB=build/mb/libsr_t.so python scripts/wgw_timing.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylerenderer_amd.op import conv as C

cdll = ctypes.CDLL(os.environ["STYLERENDERER_AMD_LIB"])
dev = "cuda"
for (b, c, n, res) in ((16, 128, 128, 256), (16, 256, 256, 128), (16, 512, 512, 64)):
    x = torch.randn(b, c, res, res, device=dev); gy = torch.randn(b, n, res, res, device=dev)
    xs = torch.randn(b, c, device=dev); gs = torch.randn(b, n, device=dev)
    for _ in range(3):
        dw = C.conv2d_wgrad_mfma(x, gy, xs, gs, 3, 1, 1, False)
    torch.cuda.synchronize()
    m = 16384
    buf = np.zeros(m * 8, dtype=np.int64)
    cdll.sr_debug_wgw_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(m * 8))
    s = buf.reshape(m, 8)
    s = s[s[:, 3] > 0]
    t = s[:, :4].astype(np.float64) * 10.0
    ph = np.diff(t, axis=1)
    mhz = s[:, 5].astype(np.float64) / ((t[:, 3] - t[:, 0]) / 1e3)
    span = (t[:, 3].max() - t[:, 0].min()) / 1e3
    strips = b * (res // 2) * (res // 16)
    print("B%d C%d N%d res%d: %d workgroups; prologue %.2f us, loop %.2f us, epilogue %.2f us; clock %.0f MHz; span %.1f us" % (
        b, c, n, res, len(s), np.median(ph[:, 0]) / 1e3, np.median(ph[:, 1]) / 1e3, np.median(ph[:, 2]) / 1e3, np.median(mhz), span))
    per = strips * (c // 64) * (n // 64) / len(s)
    print("   strips per workgroup %.1f -> %.3f us per strip (64 MFMAs per wave; %.0f cycles)" % (per, np.median(ph[:, 1]) / 1e3 / per, np.median(ph[:, 1]) / 1e3 / per * np.median(mhz)))
