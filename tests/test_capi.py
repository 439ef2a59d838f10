"""CPU: the C-ABI library builds, loads, and exports every symbol include/stylerenderer_amd.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "stylerenderer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_all_declared_symbols():
    from stylerenderer_amd import build

    path = build.build_library()
    assert os.path.isfile(path)
    handle = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(handle, n), "missing export: " + n


def test_binding_table_matches_header():
    from stylerenderer_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()
    L = _lib.lib()
    assert L.sr_abi_version() >= 1
    assert L.sr_error_string(0) == b"ok"
    assert b"invalid" in L.sr_error_string(-1)


def test_argument_validation_without_gpu():
    """Size/NULL checks happen before any launch, so they can run on the CPU container."""
    from stylerenderer_amd import _lib

    L = _lib.lib()
    assert L.sr_fused_bias_act(None, None, None, None, 3, 0, 0.2, 1.0, 16, 1, 1, 0, 0, None) == -1
    assert L.sr_fused_bias_act(None, None, None, None, 3, 0, 0.2, 1.0, 0, 1, 1, 0, 0, None) == 0
    # inconsistent out size
    assert L.sr_upfirdn2d(None, None, None, 1, 8, 8, 9, 9, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, None) == -1
    assert L.sr_rasterize_scratch_bytes(2, 10, 4, 4, 0) >= 2 * 16 * 8
    assert L.sr_rasterize_forward_f32(1, 3, 1, 0, 4, 0, 1, 0, None, None, None, None, None, 1e-6,
                                      None, 0, None, None, None, None, None) == -1
    assert L.sr_rasterize_grad_scratch_bytes(2, 10, 3, 0) >= 2 * 10 * (9 + 9) * 4 + 20
    assert L.sr_rasterize_grad_f32(1, 3, 1, 4, 4, 1, 0, None, None, 3, None, None, None, None, None, None, 0, 0,
                                   None, None, None, 1e-6, None, None) == 0    # nothing requested
    assert L.sr_rasterize_forward_cpu_f32(1, 3, 1, 4, 4, 0, 1, 0, None, None, None, None, None, 1e-6) == -1
    assert L.sr_abi_version() >= 2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from stylerenderer_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.lib()
    except _lib.NativeLibraryError as e:
        assert "no CPU/PyTorch fallback" in str(e)
    else:
        raise AssertionError("expected NativeLibraryError")
