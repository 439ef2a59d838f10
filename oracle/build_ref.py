"""TEST INFRASTRUCTURE — not part of the product.

Builds the reference's own CPU rasterizer (op/rasterize.cpp + op/rasterize.h) into
``oracle/_ref/rasterize_ref.so`` from the sources where they lie under /root/reference.
Only this repo's ``oracle/ref_rasterize_tu.cpp`` is compiled; it #includes the reference
files through ``-I /root/reference/op``.  No reference source is copied.

Runs only where /root/reference exists (the authoring container).  The built .so stays
there: oracle/_ref/ is git-ignored AND gpurun-ignored (nothing compiled from
the reference travels; the GPU box checks against the own C restatement and the fixtures).

The reference's other two native ops (fused_bias_act_kernel.cu, upfirdn2d_kernel.cu) need
CUDA headers (<cuda.h>, ATen/cuda/CUDAApplyUtils.cuh) and are therefore UNBUILDABLE here;
their oracle is the reference's Python CPU branch (see oracle/ref_shim.py).

Flags: -O2 -ffp-contract=off and no -march so x86 never fuses multiply-add; that is the
arithmetic the HIP rasterizer must match bit for bit.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("STYLERENDERER_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
NAME = "rasterize_ref"


def ref_available():
    return os.path.isfile(os.path.join(REF, "op", "rasterize.cpp"))


def built_path():
    p = os.path.join(OUT, NAME + ".so")
    return p if os.path.isfile(p) else None


def build(verbose=False):
    """Returns the path of the built module, or None when the reference is absent."""
    if not ref_available():
        return built_path()
    src = os.path.join(HERE, "ref_rasterize_tu.cpp")
    so = os.path.join(OUT, NAME + ".so")
    if os.path.isfile(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load

    load(
        name=NAME,
        sources=[src],
        extra_include_paths=[os.path.join(REF, "op")],
        extra_cflags=["-O2", "-ffp-contract=off", "-w"],
        build_directory=OUT,
        verbose=verbose,
        is_python_module=True,
    )
    return so


def load_module():
    """Imports the prebuilt pybind module (rasterize_ref.forward / .backward)."""
    so = built_path()
    if so is None:
        raise FileNotFoundError("oracle/_ref/rasterize_ref.so not built (run oracle/build_ref.py)")
    import importlib.util

    import torch  # noqa: F401  (libtorch symbols must be loaded first)

    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_capi():
    """ctypes handle on the extern "C" entry points of the same .so (z-buffer exposed)."""
    import ctypes

    import torch  # noqa: F401

    so = built_path()
    if so is None:
        raise FileNotFoundError("oracle/_ref/rasterize_ref.so not built")
    return ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)


if __name__ == "__main__":
    p = build(verbose="-v" in sys.argv)
    print("reference rasterizer:", p)
