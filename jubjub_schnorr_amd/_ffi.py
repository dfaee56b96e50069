"""ctypes binding of libjjs_gpu.so (the C ABI declared in include/jjs_gpu.h).

There is no fallback: if the HIP library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# The product library.  No environment variable can redirect the loader; tools that need another in-tree build
# of the same sources (the -DJJS_PROFILING build, A/B variants of a kernel) call select_library() explicitly
# before the first use.
LIB_PATH = os.path.join(HERE, "libjjs_gpu.so")
PROFILING_LIB_PATH = os.path.join(HERE, "libjjs_gpu_prof.so")

_P, _Z, _I = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/jjs_gpu.h
SIGNATURES = {
    "jjs_init": [_I],
    "jjs_shutdown": [],
    "jjs_last_error": [],
    "jjs_abi_version": [],
    "jjs_device_count": [],
    "jjs_collective_ranks": [],
    "jjs_verify_single": [_P, _P, _P, _P, _Z, _P, _P],
    "jjs_verify_double": [_P, _P, _P, _P, _P, _P, _Z, _P, _P],
    "jjs_verify_vargen": [_P, _P, _P, _P, _P, _Z, _P, _P],
    "jjs_verify_single_dev": [_P, _P, _P, _P, _Z, _P, _P, _P],
    "jjs_verify_double_dev": [_P, _P, _P, _P, _P, _P, _Z, _P, _P, _P],
    "jjs_verify_vargen_dev": [_P, _P, _P, _P, _P, _Z, _P, _P, _P],
    "jjs_stream_sync": [_P],
    "jjs_path_stats": [_P],
    "jjs_reserve": [_I, _I, _Z, _I],
    "jjs_trim": [],
    "jjs_memory_stats": [_P],
    "jjs_verify_single_wire_dev": [_P, _P, _P, _Z, _P, _P, _P],
    "jjs_verify_double_wire_dev": [_P, _P, _P, _Z, _P, _P, _P],
    "jjs_verify_vargen_wire_dev": [_P, _P, _P, _Z, _P, _P, _P],
    "jjs_verify_single_wire": [_P, _P, _P, _Z, _P, _P],
    "jjs_verify_double_wire": [_P, _P, _P, _Z, _P, _P],
    "jjs_verify_vargen_wire": [_P, _P, _P, _Z, _P, _P],
    "jjs_verify_single_ext_dev": [_P, _P, _P, _P, _Z, _P, _P, _P],
    "jjs_verify_double_ext_dev": [_P, _P, _P, _P, _P, _P, _Z, _P, _P, _P],
    "jjs_verify_vargen_ext_dev": [_P, _P, _P, _P, _P, _Z, _P, _P, _P],
    "jjs_verify_single_ext": [_P, _P, _P, _P, _Z, _P, _P],
    "jjs_verify_double_ext": [_P, _P, _P, _P, _P, _P, _Z, _P, _P],
    "jjs_verify_vargen_ext": [_P, _P, _P, _P, _P, _Z, _P, _P],
    "jjs_multisig_combine_dev": [_P, _P, _P, _P, _P, _P, _Z, _P, _P, _P, _P, _P, _P],
    "jjs_decompress_dev": [_P, _Z, _P, _P, _P],
    "jjs_compress_dev": [_P, _Z, _P, _P],
    "jjs_challenge_single_dev": [_P, _P, _P, _Z, _P, _P],
    "jjs_challenge_double_dev": [_P, _P, _P, _P, _P, _Z, _P, _P],
    "jjs_challenge_vargen_dev": [_P, _P, _P, _P, _Z, _P, _P],
    "jjs_sign_single_dev": [_P, _P, _P, _Z, _P, _P, _P, _P],
    "jjs_sign_double_dev": [_P, _P, _P, _Z, _P, _P, _P, _P, _P, _P],
    "jjs_sign_vargen_dev": [_P, _P, _P, _P, _Z, _P, _P, _P, _P, _P],
    "jjs_debug_fq_mul_dev": [_P, _P, _Z, _P, _P],
    "jjs_debug_poseidon_dev": [_P, _Z, _Z, _P, _P],
    "jjs_debug_point_flags_dev": [_P, _Z, _P, _P],
    "jjs_debug_half_scalars_dev": [_P, _Z, _P, _P, _P, _P],
    "jjs_debug_comb_table_bytes": [],
    "jjs_debug_comb_table": [_I, _P],
    "jjs_debug_rccl_selftest": [],
    "jjs_public_keys_dev": [_P, _Z, _P, _P, _P, _P],
}
# include/jjs_gpu_profiling.h: present in libjjs_gpu_prof.so only
PROFILING_SIGNATURES = {
    "jjs_debug_skip_phases": [ctypes.c_uint],
    "jjs_debug_allow_virtual_devices": [_I],
    "jjs_debug_force_path": [_I],
    "jjs_debug_host_timing": [_P],
    "jjs_debug_fail_key_arena": [_I],
    "jjs_debug_pin_hash_seed": [_I],
}
_RESTYPES = {"jjs_shutdown": None, "jjs_last_error": ctypes.c_char_p, "jjs_debug_comb_table_bytes": _Z}


class JjsError(RuntimeError):
    pass


_lib = None


def select_library(path: str) -> None:
    """Load another in-tree build of the engine instead of libjjs_gpu.so (profiling build, A/B kernel variants).
    Only before the first use, and only a libjjs_gpu*.so inside this package directory."""
    global LIB_PATH
    path = os.path.abspath(path)
    if _lib is not None:
        raise JjsError("select_library() must be called before the library is first used")
    if os.path.dirname(path) != HERE or not os.path.basename(path).startswith("libjjs_gpu"):
        raise JjsError(f"{path}: not an in-tree build of the engine")
    LIB_PATH = path


def lib():
    """The loaded library with typed entry points.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JjsError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # If PyTorch is going to share the process it must be loaded first: it ships its own libamdhip64
        # (same SONAME), and the dynamic loader then binds this library to that copy, so that both see
        # one HIP runtime (one current device, shared device pointers and streams).  Loaded the other
        # way round torch finds no GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the ABI and the header diverge
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, _I)
        for name, args in PROFILING_SIGNATURES.items():
            if hasattr(l, name):   # the profiling build only
                getattr(l, name).argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().jjs_last_error()
        raise JjsError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
