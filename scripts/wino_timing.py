"""Per-workgroup phase timing of k_conv_wino (debug build: scripts/build_variant.sh timing conv_wino.hip -DWINO_TIMING).
Run:  STYLERENDERER_AMD_LIB=build/mb/libsr_timing.so python scripts/wino_timing.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylerenderer_amd import _lib
from stylerenderer_amd.op.conv import conv2d_mfma

lib = _lib.lib() if hasattr(_lib, "lib") else None
cdll = ctypes.CDLL(os.environ["STYLERENDERER_AMD_LIB"])
dev = "cuda"
for (b, c, n, res) in ((16, 128, 128, 256), (16, 256, 256, 128), (16, 512, 512, 64)):
    x = torch.randn(b, c, res, res, device=dev); wt = torch.randn(9, c, n, device=dev)
    isc = torch.randn(b, c, device=dev); osc = torch.randn(b, n, device=dev)
    for _ in range(3):
        y = conv2d_mfma(x, wt, isc, osc, None, 3, 1, 1)
    torch.cuda.synchronize()
    nwg = b * (res // 32) * (res // 8) * (n // 64)
    m = min(nwg, 16384)
    buf = np.zeros(m * 8, dtype=np.int64)
    rc = cdll.sr_debug_wino_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(m * 8))
    s = buf.reshape(m, 8)
    t = s[:, :5].astype(np.float64) * 10.0          # ns (100 MHz)
    hw, xcc = s[:, 6], s[:, 7] & 0xF
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 10 * 2 + cu
    ph = np.diff(t, axis=1)
    print("rc", rc, "B%d C%d N%d res%d: %d workgroups (stamped %d), distinct CUs %d" % (b, c, n, res, nwg, m, len(np.unique(key))))
    print("  phases (us, median / mean): setup+DMA wait %.2f/%.2f  transform0 %.2f/%.2f  loop %.2f/%.2f  epilogue %.2f/%.2f  total %.2f" % (
        np.median(ph[:, 0]) / 1e3, ph[:, 0].mean() / 1e3, np.median(ph[:, 1]) / 1e3, ph[:, 1].mean() / 1e3,
        np.median(ph[:, 2]) / 1e3, ph[:, 2].mean() / 1e3, np.median(ph[:, 3]) / 1e3, ph[:, 3].mean() / 1e3,
        (t[:, 4] - t[:, 0]).mean() / 1e3))
    print("  loop per chunk %.3f us" % (np.median(ph[:, 2]) / 1e3 / (c // 8)))
    mhz = s[:, 5].astype(np.float64) / ((t[:, 4] - t[:, 0]) / 1e3)
    print("  s_memtime ticks per us (shader clock under this load, MHz): median %.0f" % np.median(mhz))
    gaps = []
    for k in np.unique(key):
        sel = np.where(key == k)[0]
        o = sel[np.argsort(t[sel, 0])]
        gaps.extend((t[o[1:], 0] - t[o[:-1], 4]).tolist())
    gaps = np.array(gaps)
    print("  turnover gap end->next start on the same CU (us): median %.2f mean %.2f p90 %.2f  (n=%d, negative %d)" % (
        np.median(gaps) / 1e3, gaps.mean() / 1e3, np.percentile(gaps, 90) / 1e3, len(gaps), int((gaps < 0).sum())))
    span = (t[:, 4].max() - t[:, 0].min()) / 1e3
    busy = (t[:, 4] - t[:, 0]).sum() / 1e3 / len(np.unique(key))
    print("  kernel span %.1f us, mean per-CU resident time %.1f us (%.1f %%)" % (span, busy, 100 * busy / span))
