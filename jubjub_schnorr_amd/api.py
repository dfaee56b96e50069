"""Host-side interface over libjjs_gpu.so.

Two levels:

* `Engine`: batch calls on torch CUDA tensors (resident data; asynchronous on torch's current
  stream) or numpy arrays (host buffers; blocking).  Arrays are uint8, SoA: scalars / field
  elements (n, 32) little-endian canonical; points (n, 64) = affine u || v.
* `PublicKey` / `Signature` & co.: the reference crate's types and method names
  (`PublicKey::verify(&self, &Signature, BlsScalar) -> Result<(), Error>`, reference
  src/keys/public.rs:114; `PublicKeyDouble::verify` src/keys/public/double.rs:86;
  `PublicKeyVarGen::verify` src/keys/public/var_gen.rs:107) plus the batch entry point a shim would
  add (`verify_batch`).  Errors mirror reference src/error.rs:13-19.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Sequence

import numpy as np

from . import _ffi

STATUS_NAMES = ("Ok", "InvalidPoint", "InvalidSignature", "Malformed")


class Error(Exception):
    """Base of the reference's `Error` enum variants reachable from verify."""


class InvalidPoint(Error):
    def __str__(self):
        return "Invalid Point"


class InvalidSignature(Error):
    def __str__(self):
        return "Invalid Signature"


class Malformed(Error):
    """Non-canonical encoding; unreachable through the Rust types (their from_bytes rejects it)."""

    def __str__(self):
        return "Malformed encoding"


_ERRORS = {1: InvalidPoint, 2: InvalidSignature, 3: Malformed}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class Engine:
    """One engine per process.  `device_count=1` (default) binds it to the current HIP device -- one process
    per GPU, what a torch.distributed rank uses; `device_count=k` drives devices 0..k-1 and 0 every visible
    device from this one process: the numpy (host-buffer) calls are then sharded across them by the library."""

    def __init__(self, device_count: int = 1):
        # When PyTorch shares the process, let it bring up the HIP runtime first: torch ships its own
        # libamdhip64 and fails with "No HIP GPUs are available" if another copy initialised the device
        # before it (observed on ROCm 7.2 / torch 2.10).
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        self._lib = _ffi.lib()
        _ffi.check(self._lib.jjs_init(int(device_count)), "jjs_init")
        self.device_count = self._lib.jjs_device_count()

    # ---- helpers ------------------------------------------------------------------------------
    @staticmethod
    def _dev_ptr(t, width, n=None):
        import torch
        if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
            raise ValueError("expected a contiguous uint8 CUDA tensor")
        if t.dim() != 2 or t.shape[1] != width or (n is not None and t.shape[0] != n):
            raise ValueError(f"expected shape (n, {width}), got {tuple(t.shape)}")
        return ctypes.c_void_p(t.data_ptr())

    @staticmethod
    def _stream():
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _host(a, width):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        if a.ndim != 2 or a.shape[1] != width:
            raise ValueError(f"expected shape (n, {width}), got {a.shape}")
        return a

    _WIDTHS = {"single": (32, 64, 64, 32), "double": (32, 64, 64, 64, 64, 32), "vargen": (32, 64, 64, 64, 32)}

    def verify_ext(self, scheme: str, *arrays, want_status: bool = True):
        """Batch verify with every point in extended coordinates, (n, 96) = U || V || Z canonical (what the Rust
        `JubJubExtended` holds; normalised on the device).  Same argument order and results as `verify`."""
        return self.verify(scheme, *arrays, want_status=want_status, _ext=True)

    def verify(self, scheme: str, *arrays, want_status: bool = True, _ext: bool = False):
        """Batch verify.  Argument order per scheme: single (u, R, PK, m); double (u, R, R', PK, PK', m);
        vargen (u, R, PK, Gen, m).  Returns (status, tally): same kind as the inputs (torch CUDA
        tensors, asynchronous on the current stream, or numpy arrays, blocking)."""
        widths = self._WIDTHS[scheme]
        suffix = ""
        if _ext:
            widths = tuple(96 if w == 64 else w for w in widths)
            suffix = "_ext"
        if len(arrays) != len(widths):
            raise ValueError(f"{scheme} verify takes {len(widths)} arrays")
        if _is_torch(arrays[0]):
            import torch
            n = arrays[0].shape[0]
            ptrs = [self._dev_ptr(a, w, n) for a, w in zip(arrays, widths)]
            dev = arrays[0].device
            status = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)[:n] if want_status else None
            tally = torch.empty(4, dtype=torch.int64, device=dev)          # the call clears it (one launch fewer than torch.zeros)
            fn = getattr(self._lib, f"jjs_verify_{scheme}{suffix}_dev")
            _ffi.check(fn(*ptrs, n, ctypes.c_void_p(status.data_ptr()) if want_status and n else None,
                          ctypes.c_void_p(tally.data_ptr()), self._stream()), f"jjs_verify_{scheme}{suffix}_dev")
            return status, tally
        host = [self._host(a, w) for a, w in zip(arrays, widths)]
        n = host[0].shape[0]
        if any(h.shape[0] != n for h in host):
            raise ValueError("all arrays must have the same number of items")
        status = np.empty(n, np.uint8)
        tally = np.zeros(4, np.uint64)
        fn = getattr(self._lib, f"jjs_verify_{scheme}{suffix}")
        _ffi.check(fn(*[h.ctypes.data_as(ctypes.c_void_p) for h in host], n, status.ctypes.data_as(ctypes.c_void_p),
                      tally.ctypes.data_as(ctypes.c_void_p)), f"jjs_verify_{scheme}{suffix}")
        return status, tally

    PATH_STAT_NAMES = ("latency", "throughput", "key_tables_wide", "key_tables_narrow", "keys_do_not_repeat",
                       "keys_probe_limit", "keys_pool_too_small", "keys_no_memory", "key_pool_bytes", "lane_launches", "lane_calls")

    def path_stats(self) -> dict:
        """Which method the calls on the current device took so far (jjs_path_stats): calls per path, and the bytes the
        per-key tables hold.  Calls still running are not in yet."""
        out = (ctypes.c_uint64 * len(self.PATH_STAT_NAMES))()
        _ffi.check(self._lib.jjs_path_stats(out), "jjs_path_stats")
        return dict(zip(self.PATH_STAT_NAMES, (int(v) for v in out)))

    _SCHEME_IDS = {"single": 0, "double": 1, "vargen": 2}
    _FORMAT_IDS = {"affine": 0, "ext": 1, "wire": 2}
    MEMORY_STAT_NAMES = ("key_pools", "slot_buffers", "host_staging", "retired")

    def reserve(self, scheme: str, n_items: int, fmt: str = "affine", host_buffers: bool = False) -> None:
        """Pre-size the engine for calls of this scheme, input format ("affine", "ext", "wire") and at most `n_items` items
        (jjs_reserve): no later call of that shape allocates.  `host_buffers`: also the staging of the numpy (blocking) calls."""
        _ffi.check(self._lib.jjs_reserve(self._SCHEME_IDS[scheme], self._FORMAT_IDS[fmt], int(n_items), int(bool(host_buffers))),
                   "jjs_reserve")

    def trim(self) -> None:
        """Wait for the device, free retired buffers and the key-table pools (jjs_trim)."""
        _ffi.check(self._lib.jjs_trim(), "jjs_trim")

    def memory_stats(self) -> dict:
        out = (ctypes.c_uint64 * len(self.MEMORY_STAT_NAMES))()
        _ffi.check(self._lib.jjs_memory_stats(out), "jjs_memory_stats")
        return dict(zip(self.MEMORY_STAT_NAMES, (int(v) for v in out)))

    _WIRE_WIDTHS = {"single": (64, 32, 32), "double": (96, 64, 32), "vargen": (64, 64, 32)}

    def verify_wire(self, scheme: str, sig, pk, m, want_status: bool = True):
        """Batch verify from the reference's wire formats (torch CUDA uint8 tensors): sig (n, 64|96|64) =
        u || R [|| R'], pk (n, 32|64|64) compressed, m (n, 32).  Points are decoded on the device; an
        undecodable item gets status 3.  Returns (status, tally), asynchronous on the current stream."""
        ws, wp, wm = self._WIRE_WIDTHS[scheme]
        if not _is_torch(sig):       # numpy: blocking host-buffer call
            hs, hp, hm = self._host(sig, ws), self._host(pk, wp), self._host(m, wm)
            n = hs.shape[0]
            status, tally = np.empty(n, np.uint8), np.zeros(4, np.uint64)
            fn = getattr(self._lib, f"jjs_verify_{scheme}_wire")
            _ffi.check(fn(*[h.ctypes.data_as(ctypes.c_void_p) for h in (hs, hp, hm)], n, status.ctypes.data_as(ctypes.c_void_p),
                          tally.ctypes.data_as(ctypes.c_void_p)), f"jjs_verify_{scheme}_wire")
            return status, tally
        import torch
        n = sig.shape[0]
        ptrs = [self._dev_ptr(sig, ws, n), self._dev_ptr(pk, wp, n), self._dev_ptr(m, wm, n)]
        status = torch.empty(max(n, 1), dtype=torch.uint8, device=sig.device)[:n] if want_status else None
        tally = torch.empty(4, dtype=torch.int64, device=sig.device)
        fn = getattr(self._lib, f"jjs_verify_{scheme}_wire_dev")
        _ffi.check(fn(*ptrs, n, ctypes.c_void_p(status.data_ptr()) if want_status and n else None,
                      ctypes.c_void_p(tally.data_ptr()), self._stream()), f"jjs_verify_{scheme}_wire_dev")
        return status, tally

    def decompress(self, enc):
        """JubJubAffine::from_bytes in bulk: (n, 32) -> ((n, 64) affine, (n,) ok)."""
        import torch
        n = enc.shape[0]
        out = torch.empty((max(n, 1), 64), dtype=torch.uint8, device=enc.device)[:n]
        ok = torch.empty(max(n, 1), dtype=torch.uint8, device=enc.device)[:n]
        _ffi.check(self._lib.jjs_decompress_dev(self._dev_ptr(enc, 32, n), n, ctypes.c_void_p(out.data_ptr()),
                                                ctypes.c_void_p(ok.data_ptr()), self._stream()), "jjs_decompress_dev")
        return out, ok

    def compress(self, pts):
        """JubJubAffine::to_bytes in bulk: (n, 64) affine -> (n, 32)."""
        import torch
        n = pts.shape[0]
        out = torch.empty((max(n, 1), 32), dtype=torch.uint8, device=pts.device)[:n]
        _ffi.check(self._lib.jjs_compress_dev(self._dev_ptr(pts, 64, n), n, ctypes.c_void_p(out.data_ptr()), self._stream()),
                   "jjs_compress_dev")
        return out

    def multisig_combine(self, z, PK, R, S, m, offsets):
        """Batch `verify_share` + `combine` + `aggregate_pk` over many transcripts (reference src/multisig.rs).
        z (N, 32), PK / R / S (N, 64), m (B, 32): torch CUDA uint8; offsets: B + 1 ints (host).  Returns
        (share_status (N,), agg_pk (B, 64), sig_u (B, 32), sig_R (B, 64), transcript_status (B,)); statuses 0 ok / 3 / 4;
        sig_u / sig_R are zero for a transcript whose status is not 0 (`combine` returns an error, not a signature)."""
        import torch
        offs = np.ascontiguousarray(offsets, dtype=np.uint32)
        B, N = len(offs) - 1, z.shape[0]
        dev_ = z.device
        new = lambda rows, w: torch.empty((max(rows, 1), w), dtype=torch.uint8, device=dev_)[:rows]  # noqa: E731
        status = torch.empty(max(N, 1), dtype=torch.uint8, device=dev_)[:N]
        tstatus = torch.empty(max(B, 1), dtype=torch.uint8, device=dev_)[:B]
        agg, su, sr = new(B, 64), new(B, 32), new(B, 64)
        o = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        _ffi.check(self._lib.jjs_multisig_combine_dev(self._dev_ptr(z, 32, N), self._dev_ptr(PK, 64, N), self._dev_ptr(R, 64, N),
                                                      self._dev_ptr(S, 64, N), self._dev_ptr(m, 32, B),
                                                      offs.ctypes.data_as(ctypes.c_void_p), B, o(status), o(tstatus), o(agg), o(su),
                                                      o(sr), self._stream()), "jjs_multisig_combine_dev")
        return status, agg, su, sr, tstatus

    def challenge(self, scheme: str, *arrays):
        """250-bit challenge per item (torch CUDA tensors): single (R, PK, m); double (R, R', PK, PK', m);
        vargen (R, PK, Gen, m)."""
        import torch
        widths = self._WIDTHS[scheme][1:]
        n = arrays[0].shape[0]
        ptrs = [self._dev_ptr(a, w, n) for a, w in zip(arrays, widths)]
        out = torch.empty((max(n, 1), 32), dtype=torch.uint8, device=arrays[0].device)[:n]
        fn = getattr(self._lib, f"jjs_challenge_{scheme}_dev")
        _ffi.check(fn(*ptrs, n, ctypes.c_void_p(out.data_ptr()), self._stream()), f"jjs_challenge_{scheme}_dev")
        return out

    def sign(self, scheme: str, sk, rnd, m, gen_scalar=None):
        """Synthetic-input generator (NOT constant time).  torch CUDA tensors in and out.
        single -> (u, R, PK); double -> (u, R, R', PK, PK'); vargen -> (u, R, PK, Gen)."""
        import torch
        n = sk.shape[0]
        dev = sk.device
        p = lambda t, w=32: self._dev_ptr(t, w, n)  # noqa: E731
        new = lambda w: torch.empty((max(n, 1), w), dtype=torch.uint8, device=dev)[:n]  # noqa: E731
        o = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        u = new(32)
        if scheme == "single":
            R, PK = new(64), new(64)
            _ffi.check(self._lib.jjs_sign_single_dev(p(sk), p(rnd), p(m), n, o(u), o(R), o(PK), self._stream()), "jjs_sign_single_dev")
            return u, R, PK
        if scheme == "double":
            R, Rp, PK, PKp = new(64), new(64), new(64), new(64)
            _ffi.check(self._lib.jjs_sign_double_dev(p(sk), p(rnd), p(m), n, o(u), o(R), o(Rp), o(PK), o(PKp), self._stream()),
                       "jjs_sign_double_dev")
            return u, R, Rp, PK, PKp
        if scheme == "vargen":
            R, PK, Gen = new(64), new(64), new(64)
            _ffi.check(self._lib.jjs_sign_vargen_dev(p(sk), p(gen_scalar), p(rnd), p(m), n, o(u), o(R), o(PK), o(Gen), self._stream()),
                       "jjs_sign_vargen_dev")
            return u, R, PK, Gen
        raise ValueError(scheme)

    def public_keys(self, sk, double: bool = False):
        """`PublicKey::from(&SecretKey)` for a batch (reference src/keys/public.rs:54-60): PK = sk*G, and with
        `double` also PK' = sk*G' (src/keys/public/double.rs:47-57).  torch CUDA tensors; NOT constant time.
        Returns (PK, bad) or (PK, PK', bad); bad[i] = 1 where sk[i] is not a canonical JubJubScalar."""
        import torch
        n = sk.shape[0]
        new = lambda w: torch.empty((max(n, 1), w), dtype=torch.uint8, device=sk.device)[:n]  # noqa: E731
        PK, PKp, bad = new(64), (new(64) if double else None), torch.empty(max(n, 1), dtype=torch.uint8, device=sk.device)[:n]
        o = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None and n else None  # noqa: E731
        _ffi.check(self._lib.jjs_public_keys_dev(self._dev_ptr(sk, 32, n), n, o(PK), o(PKp), o(bad), self._stream()),
                   "jjs_public_keys_dev")
        return (PK, PKp, bad) if double else (PK, bad)

    # ---- primitives for parity tests ------------------------------------------------------------
    def debug_fq_mul(self, a, b):
        import torch
        n = a.shape[0]
        out = torch.empty_like(a)
        _ffi.check(self._lib.jjs_debug_fq_mul_dev(self._dev_ptr(a, 32, n), self._dev_ptr(b, 32, n), n,
                                                  ctypes.c_void_p(out.data_ptr()), self._stream()), "jjs_debug_fq_mul_dev")
        return out

    def debug_poseidon(self, x):
        import torch
        n, k, w = x.shape
        assert w == 32 and x.is_contiguous()
        out = torch.empty((n, 32), dtype=torch.uint8, device=x.device)
        _ffi.check(self._lib.jjs_debug_poseidon_dev(ctypes.c_void_p(x.data_ptr()), k, n, ctypes.c_void_p(out.data_ptr()),
                                                    self._stream()), "jjs_debug_poseidon_dev")
        return out

    def debug_point_flags(self, pts):
        import torch
        n = pts.shape[0]
        out = torch.empty(n, dtype=torch.uint8, device=pts.device)
        _ffi.check(self._lib.jjs_debug_point_flags_dev(self._dev_ptr(pts, 64, n), n, ctypes.c_void_p(out.data_ptr()),
                                                       self._stream()), "jjs_debug_point_flags_dev")
        return out

    def debug_half_scalars(self, c):
        """(a, |b|, sign of b) the device derives from challenges c (n, 32): (n, 16), (n, 16), (n,) uint8 tensors."""
        import torch
        n = c.shape[0]
        a = torch.empty((max(n, 1), 16), dtype=torch.uint8, device=c.device)[:n]
        b = torch.empty((max(n, 1), 16), dtype=torch.uint8, device=c.device)[:n]
        neg = torch.empty(max(n, 1), dtype=torch.uint8, device=c.device)[:n]
        o = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        _ffi.check(self._lib.jjs_debug_half_scalars_dev(self._dev_ptr(c, 32, n), n, o(a), o(b), o(neg), self._stream()),
                   "jjs_debug_half_scalars_dev")
        return a, b, neg

    def debug_comb_table(self, which: int) -> np.ndarray:
        nbytes = self._lib.jjs_debug_comb_table_bytes()
        out = np.empty(nbytes // 4, np.uint32)
        _ffi.check(self._lib.jjs_debug_comb_table(which, out.ctypes.data_as(ctypes.c_void_p)), "jjs_debug_comb_table")
        bits = 16 if nbytes == 16 * 65536 * 112 else 8          # digits of the fixed-base comb (csrc/verify_core.h)
        return out.reshape(256 // bits, 1 << bits, 28)

    def sync(self):
        _ffi.check(self._lib.jjs_stream_sync(self._stream()), "jjs_stream_sync")


_engine = None


def engine(device_count: int = 1) -> Engine:
    global _engine
    if _engine is None:
        _engine = Engine(device_count)
    return _engine


# ------------------------------------------------------------------------------------------------
# reference-shaped types: affine points as 64 bytes, scalars as 32 bytes
# ------------------------------------------------------------------------------------------------
def _b(x, n):
    x = bytes(x)
    if len(x) != n:
        raise ValueError(f"expected {n} bytes, got {len(x)}")
    return x


@dataclass(frozen=True)
class Signature:
    """`Signature { u, R }` (reference src/signatures.rs:62-65); R as affine u || v."""
    u: bytes
    R: bytes


@dataclass(frozen=True)
class SignatureDouble:
    """`SignatureDouble { u, R, R_prime }` (reference src/signatures/double.rs:66-70)."""
    u: bytes
    R: bytes
    R_prime: bytes


@dataclass(frozen=True)
class SignatureVarGen:
    """`SignatureVarGen { u, R }` (reference src/signatures/var_gen.rs)."""
    u: bytes
    R: bytes


def _rows(items, width):
    return np.frombuffer(b"".join(_b(x, width) for x in items), np.uint8).reshape(len(items), width)


def _raise_first(status):
    for s in status:
        if s:
            raise _ERRORS[int(s)]()


@dataclass(frozen=True)
class PublicKey:
    """`PublicKey(JubJubExtended)` (reference src/keys/public.rs:52); the point as affine u || v."""
    point: bytes

    def verify(self, sig: Signature, message: bytes) -> None:
        """`PublicKey::verify` (reference src/keys/public.rs:114-135): returns None or raises."""
        _raise_first(self.verify_batch([(self, sig, message)]))

    @staticmethod
    def verify_batch(items: Sequence[tuple]) -> np.ndarray:
        """items: (PublicKey, Signature, message bytes).  Returns the status byte per item."""
        if not items:
            return np.zeros(0, np.uint8)
        st, _ = engine().verify("single", _rows([s.u for _, s, _ in items], 32), _rows([s.R for _, s, _ in items], 64),
                                _rows([k.point for k, _, _ in items], 64), _rows([m for _, _, m in items], 32))
        return st


@dataclass(frozen=True)
class PublicKeyDouble:
    """`PublicKeyDouble(pk, pk_prime)` (reference src/keys/public/double.rs:45)."""
    pk: bytes
    pk_prime: bytes

    def verify(self, sig: SignatureDouble, message: bytes) -> None:
        """`PublicKeyDouble::verify` (reference src/keys/public/double.rs:86-117)."""
        _raise_first(self.verify_batch([(self, sig, message)]))

    @staticmethod
    def verify_batch(items: Sequence[tuple]) -> np.ndarray:
        if not items:
            return np.zeros(0, np.uint8)
        st, _ = engine().verify(
            "double", _rows([s.u for _, s, _ in items], 32), _rows([s.R for _, s, _ in items], 64),
            _rows([s.R_prime for _, s, _ in items], 64), _rows([k.pk for k, _, _ in items], 64),
            _rows([k.pk_prime for k, _, _ in items], 64), _rows([m for _, _, m in items], 32))
        return st


@dataclass(frozen=True)
class PublicKeyVarGen:
    """`PublicKeyVarGen { pk, generator }` (reference src/keys/public/var_gen.rs:40-43)."""
    pk: bytes
    generator: bytes

    def verify(self, sig: SignatureVarGen, message: bytes) -> None:
        """`PublicKeyVarGen::verify` (reference src/keys/public/var_gen.rs:107-133)."""
        _raise_first(self.verify_batch([(self, sig, message)]))

    @staticmethod
    def verify_batch(items: Sequence[tuple]) -> np.ndarray:
        if not items:
            return np.zeros(0, np.uint8)
        st, _ = engine().verify(
            "vargen", _rows([s.u for _, s, _ in items], 32), _rows([s.R for _, s, _ in items], 64),
            _rows([k.pk for k, _, _ in items], 64), _rows([k.generator for k, _, _ in items], 64),
            _rows([m for _, _, m in items], 32))
        return st
