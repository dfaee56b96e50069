"""ctypes front-end of oracle/libjjs_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY.

Arrays are numpy uint8, SoA: scalars / field elements (n, 32) little-endian canonical,
affine points (n, 64) = u || v.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_lib_path = None


def build(native: bool = False) -> str:
    target = "native" if native else "all"
    subprocess.check_call(["make", "-s", "-C", HERE, target])
    return os.path.join(HERE, "libjjs_oracle_native.so" if native else "libjjs_oracle.so")


def load(native: bool = False):
    """Load the library, building it when missing (gcc is in the image)."""
    global _lib, _lib_path
    path = os.path.join(HERE, "libjjs_oracle_native.so" if native else "libjjs_oracle.so")
    if _lib is not None and _lib_path == path:
        return _lib
    if not os.path.exists(path):
        build(native)
    lib = ctypes.CDLL(path)
    P, Z, I = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    sigs = {
        "jjo_max_threads": [],
        "jjo_verify_single": [P, P, P, P, Z, P, P, I],
        "jjo_verify_double": [P, P, P, P, P, P, Z, P, P, I],
        "jjo_verify_vargen": [P, P, P, P, P, Z, P, P, I],
        "jjo_sign_single": [P, P, P, Z, P, P, P, I],
        "jjo_sign_double": [P, P, P, Z, P, P, P, P, P, I],
        "jjo_sign_vargen": [P, P, P, P, Z, P, P, P, P, I],
        "jjo_fq_mul": [P, P, Z, P],
        "jjo_poseidon": [P, Z, Z, P, I],
        "jjo_poseidon_tagged": [P, Z, Z, P, P, I],
        "jjo_scalar_mul": [P, P, Z, P, I],
        "jjo_point_flags": [P, Z, P, I],
        "jjo_point_add": [P, P, Z, P],
        "jjo_multisig_combine": [P, P, P, P, P, P, Z, P, P, P, P, P, P, I],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = I
    _lib, _lib_path = lib, path
    return lib


def _c(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2 and a.shape[1] == width, (a.shape, width)
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def max_threads(native=False) -> int:
    return load(native).jjo_max_threads()


def verify_single(u, R, PK, m, threads=0, want_c=False, native=False):
    u, R, PK, m = _c(u, 32), _c(R, 64), _c(PK, 64), _c(m, 32)
    n = len(u)
    st = np.empty(n, np.uint8)
    c = np.zeros((n, 32), np.uint8) if want_c else None
    rc = load(native).jjo_verify_single(_p(u), _p(R), _p(PK), _p(m), n, _p(st), _p(c) if want_c else None, threads)
    assert rc == 0
    return (st, c) if want_c else st


def verify_double(u, R, Rp, PK, PKp, m, threads=0, want_c=False, native=False):
    u, R, Rp, PK, PKp, m = _c(u, 32), _c(R, 64), _c(Rp, 64), _c(PK, 64), _c(PKp, 64), _c(m, 32)
    n = len(u)
    st = np.empty(n, np.uint8)
    c = np.zeros((n, 32), np.uint8) if want_c else None
    rc = load(native).jjo_verify_double(_p(u), _p(R), _p(Rp), _p(PK), _p(PKp), _p(m), n, _p(st),
                                        _p(c) if want_c else None, threads)
    assert rc == 0
    return (st, c) if want_c else st


def verify_vargen(u, R, PK, Gen, m, threads=0, want_c=False, native=False):
    u, R, PK, Gen, m = _c(u, 32), _c(R, 64), _c(PK, 64), _c(Gen, 64), _c(m, 32)
    n = len(u)
    st = np.empty(n, np.uint8)
    c = np.zeros((n, 32), np.uint8) if want_c else None
    rc = load(native).jjo_verify_vargen(_p(u), _p(R), _p(PK), _p(Gen), _p(m), n, _p(st),
                                        _p(c) if want_c else None, threads)
    assert rc == 0
    return (st, c) if want_c else st


def sign_single(sk, rnd, m, threads=0):
    sk, rnd, m = _c(sk, 32), _c(rnd, 32), _c(m, 32)
    n = len(sk)
    u, R, PK = np.empty((n, 32), np.uint8), np.empty((n, 64), np.uint8), np.empty((n, 64), np.uint8)
    rc = load().jjo_sign_single(_p(sk), _p(rnd), _p(m), n, _p(u), _p(R), _p(PK), threads)
    assert rc == 0
    return u, R, PK


def sign_double(sk, rnd, m, threads=0):
    sk, rnd, m = _c(sk, 32), _c(rnd, 32), _c(m, 32)
    n = len(sk)
    u = np.empty((n, 32), np.uint8)
    R, Rp, PK, PKp = (np.empty((n, 64), np.uint8) for _ in range(4))
    rc = load().jjo_sign_double(_p(sk), _p(rnd), _p(m), n, _p(u), _p(R), _p(Rp), _p(PK), _p(PKp), threads)
    assert rc == 0
    return u, R, Rp, PK, PKp


def sign_vargen(sk, g, rnd, m, threads=0):
    sk, g, rnd, m = _c(sk, 32), _c(g, 32), _c(rnd, 32), _c(m, 32)
    n = len(sk)
    u = np.empty((n, 32), np.uint8)
    R, PK, Gen = (np.empty((n, 64), np.uint8) for _ in range(3))
    rc = load().jjo_sign_vargen(_p(sk), _p(g), _p(rnd), _p(m), n, _p(u), _p(R), _p(PK), _p(Gen), threads)
    assert rc == 0
    return u, R, PK, Gen


def fq_mul(a, b):
    a, b = _c(a, 32), _c(b, 32)
    out = np.empty_like(a)
    assert load().jjo_fq_mul(_p(a), _p(b), len(a), _p(out)) == 0
    return out


def poseidon(inputs, threads=0):
    """inputs (n, k, 32) -> (n, 32) untruncated digests."""
    inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
    n, k, w = inputs.shape
    assert w == 32
    out = np.empty((n, 32), np.uint8)
    assert load().jjo_poseidon(_p(inputs), k, n, _p(out), threads) == 0
    return out


def poseidon_any(inputs, threads=0):
    """inputs (n, k, 32), any k >= 1 -> (n, 32) untruncated digests; the SAFE tag of k comes from the Python oracle."""
    import jjs_oracle as o
    inputs = np.ascontiguousarray(inputs, dtype=np.uint8)
    n, k, w = inputs.shape
    assert w == 32
    tag = np.frombuffer(o.sponge_tag(k).to_bytes(32, "little"), np.uint8).copy()
    out = np.empty((n, 32), np.uint8)
    assert load().jjo_poseidon_tagged(_p(inputs), k, n, _p(tag), _p(out), threads) == 0
    return out


def scalar_mul(P, k, threads=0):
    P, k = _c(P, 64), _c(k, 32)
    out = np.empty_like(P)
    assert load().jjo_scalar_mul(_p(P), _p(k), len(P), _p(out), threads) == 0
    return out


def point_flags(P, threads=0):
    P = _c(P, 64)
    out = np.empty(len(P), np.uint8)
    assert load().jjo_point_flags(_p(P), len(P), _p(out), threads) == 0
    return out


def point_add(P, Q):
    P, Q = _c(P, 64), _c(Q, 64)
    out = np.empty_like(P)
    assert load().jjo_point_add(_p(P), _p(Q), len(P), _p(out)) == 0
    return out


def multisig_combine(z, PK, R, S, m, offsets, threads=0, native=False):
    """combine / verify_share / aggregate_pk over many transcripts with the reference's algorithm (jjo_multisig_combine):
    returns (share_status (N,), transcript_status (B,), agg_pk (B, 64), sig_u (B, 32), sig_R (B, 64))."""
    import jjs_oracle as o
    z, PK, R, S, m = _c(z, 32), _c(PK, 64), _c(R, 64), _c(S, 64), _c(m, 32)
    offs = np.ascontiguousarray(offsets, dtype=np.uint32)
    B, N = len(offs) - 1, len(z)
    tags = np.zeros((max(B, 1), 2, 32), np.uint8)
    cache = {}
    for t in range(B):
        n = int(offs[t + 1]) - int(offs[t])
        if n not in cache:
            cache[n] = [np.frombuffer(o.sponge_tag(k).to_bytes(32, "little"), np.uint8) for k in (2 + 2 * n, 3 + 4 * n)]
        tags[t, 0], tags[t, 1] = cache[n]
    share = np.zeros(max(N, 1), np.uint8)[:N]
    tst = np.zeros(max(B, 1), np.uint8)[:B]
    agg, su, sr = np.zeros((max(B, 1), 64), np.uint8)[:B], np.zeros((max(B, 1), 32), np.uint8)[:B], np.zeros((max(B, 1), 64), np.uint8)[:B]
    rc = load(native).jjo_multisig_combine(_p(z), _p(PK), _p(R), _p(S), _p(m), offs.ctypes.data_as(ctypes.c_void_p), B, _p(tags), _p(share), _p(tst), _p(agg), _p(su), _p(sr), threads)
    assert rc == 0
    return share, tst, agg, su, sr
