"""The C-ABI library loads on a CPU-only box and exports every symbol include/jjs_gpu.h declares.
No compute calls here (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "jjs_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jjs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    for s in ("jjs_init", "jjs_shutdown", "jjs_verify_single", "jjs_verify_double", "jjs_verify_vargen",
              "jjs_verify_single_dev", "jjs_verify_double_dev", "jjs_verify_vargen_dev", "jjs_challenge_single_dev",
              "jjs_sign_single_dev"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from jubjub_schnorr_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        pytest.fail(f"{_ffi.LIB_PATH} missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), s
    assert lib.jjs_abi_version() == 2


def test_python_binding_covers_the_header():
    from jubjub_schnorr_amd import _ffi
    assert sorted(_ffi.SIGNATURES) == header_symbols()
    _ffi.lib()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from jubjub_schnorr_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_ffi.JjsError):
        _ffi.lib()


def test_calls_before_init_report_not_initialised():
    from jubjub_schnorr_amd import _ffi
    lib = _ffi.lib()
    # fresh process state is not guaranteed (other tests may have initialised), so only check the
    # contract when the engine is down
    lib.jjs_shutdown()
    assert lib.jjs_stream_sync(None) == -4
    assert b"jjs_init" in lib.jjs_last_error()
