"""The C-ABI library loads on a CPU-only box and exports every symbol include/jjs_gpu.h declares.
No compute calls here (no GPU)."""
import ctypes
import os
import re
import subprocess
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="jjs_gpu.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
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
    assert lib.jjs_abi_version() == 5


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_product_library_has_no_bypass_switch():
    """The ablation switches and the logical-device mode are compiled out of libjjs_gpu.so; they exist only in
    the -DJJS_PROFILING build, which declares them in include/jjs_gpu_profiling.h."""
    from jubjub_schnorr_amd import _ffi
    prof_syms = header_symbols("jjs_gpu_profiling.h")
    assert prof_syms == ["jjs_debug_allow_virtual_devices", "jjs_debug_fail_key_arena", "jjs_debug_force_path", "jjs_debug_host_timing",
                         "jjs_debug_pin_hash_seed", "jjs_debug_skip_phases"] == sorted(_ffi.PROFILING_SIGNATURES)
    product = exported(os.path.join(ROOT, "jubjub_schnorr_amd", "libjjs_gpu.so"))
    for s in prof_syms:
        assert s not in product, s
    assert not any("skip" in s or "virtual" in s or "force" in s or "fail" in s or "pin_" in s for s in product if s.startswith("jjs_"))
    # no string of the product binary names an environment switch
    blob = open(os.path.join(ROOT, "jubjub_schnorr_amd", "libjjs_gpu.so"), "rb").read()
    assert b"JJS_DEBUG" not in blob and b"JJS_GPU_LIB" not in blob
    prof = exported(_ffi.PROFILING_LIB_PATH)
    for s in prof_syms + header_symbols():
        assert s in prof, s


def test_loader_ignores_the_environment(monkeypatch, tmp_path):
    monkeypatch.setenv("JJS_GPU_LIB", str(tmp_path / "evil.so"))
    import importlib
    from jubjub_schnorr_amd import _ffi
    fresh = importlib.reload(_ffi)
    try:
        assert fresh.LIB_PATH == os.path.join(ROOT, "jubjub_schnorr_amd", "libjjs_gpu.so")
        with pytest.raises(fresh.JjsError):
            fresh.select_library(str(tmp_path / "libjjs_gpu_evil.so"))       # not inside the package
    finally:
        monkeypatch.delenv("JJS_GPU_LIB")
        importlib.reload(_ffi)


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


def test_last_error_is_per_thread():
    """Two host threads provoking different argument errors at once: each reads its own message, and a pointer
    obtained earlier is not rewritten by the other thread (include/jjs_gpu.h: threading)."""
    from jubjub_schnorr_amd import _ffi
    lib = _ffi.lib()
    lib.jjs_shutdown()
    errors = []

    def worker(bad_count, needle):
        try:
            for _ in range(2000):
                assert lib.jjs_init(bad_count) == -1
                msg = lib.jjs_last_error()
                assert needle in msg, (needle, msg)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(-7, b"got -7")), threading.Thread(target=worker, args=(99, b"got 99")),
          threading.Thread(target=worker, args=(-3, b"got -3"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:1]
    # a thread that never failed sees an empty message
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.jjs_last_error()))
    t.start(); t.join()
    assert seen == [b""]
