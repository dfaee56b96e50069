"""ctypes loader for tests/hostbuild/libjjs_hosttest.so: the product's csrc/*.h compiled for the CPU."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "hostbuild", "host_harness.cpp")
LIB = os.path.join(HERE, "hostbuild", "libjjs_hosttest.so")
CSRC = os.path.join(ROOT, "jubjub_schnorr_amd", "csrc")
_lib = None


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    return any(os.path.getmtime(d) > t for d in deps)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if _stale():
        # JJS_HOST_SANITIZE=1 python -m pytest tests/test_hostbuild.py  -> the same tests under UBSan
        san = ["-O1", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=undefined"] if os.environ.get("JJS_HOST_SANITIZE") else ["-O2"]
        subprocess.check_call(["g++", *san, "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas",
                               "-I" + CSRC, "-o", LIB, SRC])
    _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def fq_mul(a, b):
    a, b = _c(a), _c(b); out = np.empty_like(a)
    load().jjs_host_fq_mul(_p(a), _p(b), ctypes.c_size_t(len(a)), _p(out)); return out


def fq_sqr(a):
    a = _c(a); out = np.empty_like(a)
    load().jjs_host_fq_sqr(_p(a), ctypes.c_size_t(len(a)), _p(out)); return out


def fq_inv(a):
    a = _c(a); out = np.empty_like(a)
    load().jjs_host_fq_inv(_p(a), ctypes.c_size_t(len(a)), _p(out)); return out


def fq_addsub(a, b):
    a, b = _c(a), _c(b); out = np.empty((len(a), 64), np.uint8)
    load().jjs_host_fq_addsub(_p(a), _p(b), ctypes.c_size_t(len(a)), _p(out)); return out[:, :32], out[:, 32:]


def poseidon(x):
    x = _c(x); n, k, _ = x.shape; out = np.empty((n, 32), np.uint8)
    load().jjs_host_poseidon(_p(x), ctypes.c_size_t(k), ctypes.c_size_t(n), _p(out)); return out


def point_flags(P):
    P = _c(P); out = np.empty(len(P), np.uint8)
    load().jjs_host_point_flags(_p(P), ctypes.c_size_t(len(P)), _p(out)); return out


def comb_bits():
    return load().jjs_host_comb_bits()


def comb_entry_matches_device_builder(which, i, b):
    return bool(load().jjs_host_comb_entry_matches_device_builder(which, i, b))


def comb_entry(which, i, b):
    out = np.empty(96, np.uint8)
    load().jjs_host_comb_entry(which, i, b, _p(out)); return out


def set_split_prepare(on):
    """First pass as PREP_HEAD + PREP_TAIL (what the device launches while a batch's keys are being counted)."""
    load().jjs_host_set_split_prepare(int(bool(on)))


def verify(scheme, b, want_c=False):
    from helpers import ARG_ORDER
    args = [_c(b[k]) for k in ARG_ORDER[scheme]]
    n = len(args[0])
    st = np.empty(n, np.uint8); tally = np.zeros(4, np.uint64)
    c = np.zeros((n, 32), np.uint8) if want_c else None
    fn = getattr(load(), "jjs_host_verify_" + scheme)
    fn(*[_p(a) for a in args], ctypes.c_size_t(n), _p(st), _p(tally), _p(c))
    return (st, tally, c) if want_c else (st, tally)


def half_size(c):
    c = _c(c); out = np.empty((len(c), 33), np.uint8)
    load().jjs_host_half_size(_p(c), ctypes.c_size_t(len(c)), _p(out)); return out


def decompress(c):
    c = _c(c); out = np.empty((len(c), 64), np.uint8); ok = np.empty(len(c), np.uint8)
    load().jjs_host_decompress(_p(c), ctypes.c_size_t(len(c)), _p(out), _p(ok)); return out, ok


def multisig(z, PK, R, S, m, offsets):
    z, PK, R, S, m = _c(z), _c(PK), _c(R), _c(S), _c(m)
    offs = np.ascontiguousarray(offsets, dtype=np.uint32)
    B, N = len(offs) - 1, len(z)
    status = np.empty(N, np.uint8); agg = np.empty((B, 64), np.uint8); su = np.empty((B, 32), np.uint8); sr = np.empty((B, 64), np.uint8)
    ts = np.empty(B, np.uint8)
    load().jjs_host_multisig(_p(z), _p(PK), _p(R), _p(S), _p(m), _p(offs), ctypes.c_size_t(B), _p(status), _p(agg), _p(su), _p(sr), _p(ts))
    return status, agg, su, sr, ts


def safe_tag(n_inputs: int, table: bool = False):
    out = np.zeros(9, np.uint32)
    fn = load().jjs_host_safe_tag_table if table else load().jjs_host_safe_tag
    assert fn(ctypes.c_uint32(n_inputs), _p(out)) == 0
    return out


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def raw_mul(a, b):
    a, b = _u32(a), _u32(b); out = np.empty_like(a)
    load().jjs_host_raw_mul(_p(a), _p(b), ctypes.c_size_t(len(a)), _p(out)); return out


def raw_sqr(a):
    a = _u32(a); out = np.empty_like(a)
    load().jjs_host_raw_sqr(_p(a), ctypes.c_size_t(len(a)), _p(out)); return out


def raw_dot5(t, row):
    t = _u32(t); n = len(t); d = np.empty((n, 9), np.uint32); s = np.empty((n, 9), np.uint32)
    load().jjs_host_raw_dot5(_p(t), row, ctypes.c_size_t(n), _p(d), _p(s)); return d, s


def normalize(ext_arrays, lanes=1):
    """extended (n, 96) arrays -> affine (n, 64) arrays + malformed flags, through csrc/normalize.h on the CPU."""
    ext = [np.ascontiguousarray(a, dtype=np.uint8) for a in ext_arrays]
    n, k = len(ext[0]), len(ext)
    outs = [np.zeros((n, 64), np.uint8) for _ in ext]
    bad = np.zeros(n, np.uint8)
    PP = ctypes.c_void_p * k
    load().jjs_host_normalize(PP(*[a.ctypes.data for a in ext]), k, ctypes.c_size_t(n), ctypes.c_size_t(lanes),
                              PP(*[a.ctypes.data for a in outs]), _p(bad))
    return outs, bad


def verify_small(scheme, b, positions=4):
    """The latency path (csrc/small_batch.h) on the CPU build: single and double schemes, 4 or 8 pieces."""
    from helpers import ARG_ORDER
    args = [_c(b[k]) for k in ARG_ORDER[scheme]]
    n = len(args[0])
    st = np.empty(n, np.uint8); tally = np.zeros(4, np.uint64)
    fn = getattr(load(), "jjs_host_verify_small_" + scheme)
    fn(*[_p(a) for a in args], ctypes.c_size_t(n), _p(st), _p(tally), int(positions))
    return st, tally


def verify_keyed(scheme, b, window=5):
    """The key-table path (csrc/key_tables.h) on the CPU build, with 5- or 6-bit windows."""
    from helpers import ARG_ORDER
    args = [_c(b[k]) for k in ARG_ORDER[scheme]]
    n = len(args[0])
    st = np.empty(n, np.uint8); tally = np.zeros(4, np.uint64)
    assert load().jjs_host_set_key_window(int(window)) == 0
    fn = getattr(load(), "jjs_host_verify_keyed_" + scheme)
    fn(*[_p(a) for a in args], ctypes.c_size_t(n), _p(st), _p(tally))
    return st, tally
