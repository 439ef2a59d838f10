"""Ahead-of-time build of libstylerenderer_hip.so (gfx950 only, hipcc cross-compiles without a GPU).

The library is built IN-TREE next to this file so that it travels with the repository snapshot
to the GPU box.  Per-file flags:
  * the element-wise / FIR / rasterizer kernels are compiled with -ffp-contract=off and
    correctly-rounded fp32 division: their results are compared bit for bit with the CPU oracle;
  * the MFMA convolution kernels use default contraction (the matrix core is an fma chain anyway).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstylerenderer_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
          "-fhip-fp32-correctly-rounded-divide-sqrt"]
EXACT = ["-ffp-contract=off"]
# no SLP packing (v_pk_mul / v_pk_add on neighbouring accumulators) for the FIR kernels: it costs the plain blur 73
# register shuffles and 121 instead of 64 registers (four instead of eight waves per SIMD; 3.1 -> 3.8-4.2 TB/s
# without it).  The arithmetic is the same IEEE operations either way.  Measured on the other streaming files too:
# k_smallconv_dw +8 %, k_smallconv_fwd / dx at 64^2 -20..30 %, whole step slower — left alone there.
NOSLP = ["-fno-slp-vectorize"]

SOURCES = [
    ("capi.hip", EXACT),
    ("fused_bias_act.hip", EXACT),
    ("upfirdn2d.hip", EXACT + NOSLP),
    ("rasterize.hip", EXACT),
    ("fused_elem.hip", EXACT),
    ("weight_prep.hip", EXACT),
    ("style_linear.hip", EXACT),
    ("bank_mm.hip", EXACT),
    ("mesh.hip", EXACT),
    ("lpips.hip", EXACT),
    ("conv_mfma.hip", []),
    ("conv_wino.hip", []),
    ("conv_wgrad_wino.hip", []),
    ("conv_wgrad_mfma.hip", []),
    ("conv_wgrad_bf16x3.hip", []),
    ("conv_s2_bf16x3.hip", []),
    ("conv_generic.hip", []),
    ("conv1x1_gemm.hip", []),
]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps(src):
    deps = [src, os.path.join(CSRC, "common.h"),
            os.path.join(HERE, "..", "include", "stylerenderer_amd.h"), os.path.abspath(__file__)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    objs, rebuilt = [], False
    procs = []
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < _deps(src):
            cmd = [hipcc] + COMMON + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on %s" % name)
        if verbose and out:
            sys.stdout.write(out.decode(errors="replace"))
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="-f" in sys.argv, verbose=True))
