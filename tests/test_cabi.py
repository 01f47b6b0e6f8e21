"""CPU: the C-ABI shared library loads and exports every symbol include/nesr_hip.h declares;
the ctypes binding lists the same set.  No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "nesr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nesr_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from neural_enhanced_super_resolution_amd import _lib
    return _lib.load()


def test_header_declares_the_documented_entry_points():
    syms = header_symbols()
    for s in ("nesr_create", "nesr_load_weight", "nesr_finalize_weights", "nesr_forward", "nesr_forward_u8",
              "nesr_workspace_bytes", "nesr_destroy", "nesr_last_error"):
        assert s in syms


def test_library_exports_every_header_symbol(lib):
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/nesr_hip.h but not exported"


def test_binding_covers_exactly_the_header():
    from neural_enhanced_super_resolution_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.nesr_version()
    assert isinstance(lib.nesr_last_error(), bytes)


def test_create_rejects_bad_arguments_without_touching_a_device(lib):
    h = ctypes.c_void_p()
    assert lib.nesr_create(ctypes.byref(h), 0, 12, 3, 64, 23, 32, 3, 0) < 0        # unshuffle 3
    assert b"unshuffle" in lib.nesr_last_error()
    assert lib.nesr_create(ctypes.byref(h), 0, 12, 2, 48, 23, 32, 3, 0) < 0        # num_feat 48
    assert lib.nesr_create(ctypes.byref(h), 0, 12, 2, 64, 23, 32, 3, 7) < 0        # dtype 7
    assert h.value is None


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from neural_enhanced_super_resolution_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libnesr_hip.so"))
    with pytest.raises(_lib.NesrHipError, match="no CPU"):
        _lib.load()
