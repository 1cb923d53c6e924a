"""The C-ABI library loads without a GPU and exports every symbol include/indicasr.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "indicasr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ia_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from indic_cl_asr_amd import _lib
    names = _declared()
    assert len(names) >= 15
    so = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), f"{n} declared in indicasr.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert set(_lib.SIGNATURES) <= set(names)
    assert _lib.version().startswith("indicasr-hip gfx950")


def test_pure_host_entry_points():
    from indic_cl_asr_amd import _lib
    L = _lib.lib()
    assert L.ia_rnnt_workspace_bytes(32, 376, 106) > 0
    assert L.ia_rnnt_workspace_bytes(1, 10, 2000) == 0          # U1 > 1024 unsupported
    assert L.ia_joint_ld(257) == 264 and L.ia_cl_chunk_elems() == 4096
    # argument validation happens before any device work: null pointers -> IA_INVALID_VALUE (-1)
    assert L.ia_rnnt_loss(None, None, None, None, 1, 1, 1, 4, 0, 0.0, 0.0, None, None, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from indic_cl_asr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.lib()
        raise AssertionError("expected RuntimeError")
    except RuntimeError as e:
        assert "no fallback" in str(e)
