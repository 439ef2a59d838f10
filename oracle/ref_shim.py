"""TEST INFRASTRUCTURE — not part of the product; works only where /root/reference exists.

Imports the reference's Python hot path (op/, layers.py, model.py) in the authoring
container so that golden vectors can be generated from it (oracle/make_golden.py) and so
the CPU restatements in oracle/ can be validated against it.  Nothing here travels to the
GPU box in a usable form: /root/reference is absent there and every function raises.

The reference cannot be imported as-is on a CUDA-less box:
  * op/fused_act.py:11, op/upfirdn2d.py:10, op/rasterize.py:10 JIT-build CUDA extensions at
    import time -> ``torch.utils.cpp_extension.load`` is replaced while importing ``op``;
    the fused / upfirdn2d stubs are never touched for CPU tensors (op/fused_act.py:87,
    op/upfirdn2d.py:146); ``rasterize`` resolves to oracle/_ref/rasterize_ref.so, which is
    the reference's own op/rasterize.cpp compiled from where it lies (oracle/build_ref.py).
  * layers.py uses ``math`` without importing it (layers.py:210) -> injected.
  * ConvLayer(activate=False) calls ``.lower()`` on a bool (layers.py:357) -> mapped to
    'none' (skip conv: no bias, no activation), needed for ResBlock / GeneratorWithMap /
    Discriminator only.
"""
import math
import os
import sys
import types

REF = os.environ.get("STYLERENDERER_REFERENCE", "/root/reference")
_cache = {}


def available():
    return os.path.isfile(os.path.join(REF, "model.py"))


def load():
    """Returns a namespace with .op, .layers, .model (reference modules)."""
    if "ns" in _cache:
        return _cache["ns"]
    if not available():
        raise RuntimeError("reference not present at %s" % REF)
    import torch.utils.cpp_extension as ce

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import build_ref

    build_ref.build()
    real_load = ce.load

    def fake_load(name, sources=None, **kw):
        if name == "rasterize":
            return build_ref.load_module()
        return types.SimpleNamespace()

    ce.load = fake_load
    sys.path.insert(0, REF)
    try:
        import op as ref_op  # noqa: E402
        import layers as ref_layers  # noqa: E402

        ref_layers.math = math
        orig_init = ref_layers.ConvLayer.__init__

        def patched_init(self, in_channel, out_channel, kernel_size, downsample=False,
                         blur_kernel=[1, 3, 3, 1], bias=True, activate="lrelu"):
            if activate is False:
                activate = "none"
            orig_init(self, in_channel, out_channel, kernel_size, downsample=downsample,
                      blur_kernel=blur_kernel, bias=bias, activate=activate)

        ref_layers.ConvLayer.__init__ = patched_init
        import model as ref_model  # noqa: E402
    finally:
        ce.load = real_load
        sys.path.remove(REF)
    ns = types.SimpleNamespace(op=ref_op, layers=ref_layers, model=ref_model,
                               rasterize_op=build_ref.load_module())
    _cache["ns"] = ns
    return ns


if __name__ == "__main__":
    import torch

    ns = load()
    g = ns.model.Generator(64, 512, 8)
    print("Generator(64) params:", sum(p.numel() for p in g.parameters()))
    img, _ = g([torch.randn(2, 512)])
    print("image", tuple(img.shape), float(img.abs().mean()))
