"""CPU oracle (pure-Python big integers) for Schnorr-on-JubJub verification.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``jubjub_schnorr_amd/``,
``include/``, the HIP library) may import, link or execute this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and only as the checker.

What it restates (reference = /root/reference, crate jubjub-schnorr v0.7.0-rc.0):

* ``PublicKey::verify``           src/keys/public.rs:114-135, ``is_valid`` :159-164
* ``Signature::is_valid``         src/signatures.rs:93-98
* ``challenge_hash`` (single)     src/signatures.rs:122-140
* ``PublicKeyDouble::verify``     src/keys/public/double.rs:86-117, ``is_valid`` :145-157
* ``SignatureDouble::is_valid``   src/signatures/double.rs:108-119
* ``challenge_hash`` (double)     src/signatures/double.rs:151-177, tag :24-25
* ``PublicKeyVarGen::verify``     src/keys/public/var_gen.rs:107-133, ``is_valid`` :160-172
* ``challenge_hash`` (var-gen)    src/signatures/var_gen.rs:121-142
* ``Error`` classes               src/error.rs:13-19
* signing (input generation only) src/keys/secret.rs:174-194, src/keys/secret/double.rs:56-85,
                                  src/keys/secret/var_gen.rs:162-172,228-256, src/nonce.rs:26-107
* multisig transcript (KAT only)  src/multisig.rs:393-500

The field / curve / Poseidon arithmetic lives in third-party crates that are NOT in
/root/reference (dusk-bls12_381 0.14, dusk-jubjub 0.15, dusk-poseidon 0.42.0-rc.0,
dusk-safe; Cargo.toml:22-30).  Their published algorithms are restated here and the
restatement is PINNED by the reference's own byte-level known-answer vectors
(src/multisig.rs:544-735 and tests/serde.rs:34-142), see tests/test_oracle_kat.py.

Status codes (mirrors src/error.rs:17-19 and the precedence at src/keys/public.rs:119-132):
    0 = Ok, 1 = InvalidPoint, 2 = InvalidSignature, 3 = Malformed (non-canonical bytes;
    cannot occur through the Rust types, defined so the byte-level ABI is total).
"""
from __future__ import annotations

import hashlib

# ----------------------------------------------------------------------------
# A.1 fields
# ----------------------------------------------------------------------------
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # BlsScalar modulus
R_ORDER = 0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7  # JubJubScalar modulus
MONT_R = (1 << 256) % Q

OK, INVALID_POINT, INVALID_SIGNATURE, MALFORMED = 0, 1, 2, 3


def fq_inv(a: int) -> int:
    return pow(a, Q - 2, Q)


def le32(x: int) -> bytes:
    return int(x).to_bytes(32, "little")


def from_le(b: bytes) -> int:
    return int.from_bytes(b, "little")


# ----------------------------------------------------------------------------
# A.2 JubJub: -u^2 + v^2 = 1 + d u^2 v^2 over Fq
# ----------------------------------------------------------------------------
D = (-(10240 * fq_inv(10241))) % Q
assert D == 0x2A9318E74BFA2B48F5FD9207E6BD7FD4292D7F6D37579D2601065FD6D6343EB1

G = (0x3FD2814C43AC65A6F1FBF02D0FD6CCE62E3EBB21FD6C54ED4DF7B7FFEC7BEACA, 0x12)
G_NUMS = (
    0x5E67B8F316F414F7BD9514C773FD4456931E316A39FE4541921710179DF76377,
    0x43D80EB3B2F3EB1B7B162DBEEB3B34FD9949BA0F82A5507A6705B707162E3EF8,
)
IDENTITY = (0, 1)


def is_on_curve(p) -> bool:
    u, v = p
    u2, v2 = u * u % Q, v * v % Q
    return (v2 - u2) % Q == (1 + D * u2 % Q * v2) % Q


def is_identity(p) -> bool:
    return p[0] % Q == 0 and p[1] % Q == 1


# extended coordinates (U, V, Z, T) with T = UV/Z; unified a=-1 formulas, complete
# on the whole curve because d is a non-square.
def _ext(p):
    u, v = p
    return (u % Q, v % Q, 1, u * v % Q)


def _ext_add(p1, p2):
    u1, v1, z1, t1 = p1
    u2, v2, z2, t2 = p2
    a = (v1 - u1) * (v2 - u2) % Q
    b = (v1 + u1) * (v2 + u2) % Q
    c = 2 * D * t1 % Q * t2 % Q
    d = 2 * z1 * z2 % Q
    e, f, g, h = (b - a) % Q, (d - c) % Q, (d + c) % Q, (b + a) % Q
    return (e * f % Q, g * h % Q, f * g % Q, e * h % Q)


def _ext_to_affine(p):
    u, v, z, _ = p
    zi = fq_inv(z)
    return (u * zi % Q, v * zi % Q)


def add(p1, p2):
    """Affine group law (complete).  Defined for on-curve points."""
    return _ext_to_affine(_ext_add(_ext(p1), _ext(p2)))


def neg(p):
    return ((-p[0]) % Q, p[1] % Q)


def mul_ext(p, k: int, nbits: int = 256):
    """MSB-first double-and-add (the reference's algorithm, A.2), extended result.

    Like the dependency crate the loop is formula-driven: it runs the same unified
    formulas on whatever coordinates it is given, on-curve or not.
    """
    acc = _ext(IDENTITY)
    base = _ext(p)
    for i in reversed(range(nbits)):
        acc = _ext_add(acc, acc)
        if (k >> i) & 1:
            acc = _ext_add(acc, base)
    return acc


def mul(p, k: int):
    return _ext_to_affine(mul_ext(p, k))


def is_torsion_free(p) -> bool:
    """[r]P == identity (A.2)."""
    u, v, z, _ = mul_ext(p, R_ORDER)
    return u % Q == 0 and (v - z) % Q == 0


def point_is_valid(p) -> bool:
    """torsion-free AND on-curve AND not identity (src/keys/public.rs:159-164)."""
    if not is_on_curve(p):
        return False
    if is_identity(p):
        return False
    return is_torsion_free(p)


def compress(p) -> bytes:
    u, v = p
    b = bytearray(le32(v))
    b[31] |= (u & 1) << 7
    return bytes(b)


def fq_sqrt(a: int):
    """Tonelli-Shanks in Fq (2-adicity 32).  Returns a root or None."""
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        return None
    s, t = 32, (Q - 1) >> 32
    z = 7
    while pow(z, (Q - 1) // 2, Q) == 1:
        z += 1
    c = pow(z, t, Q)
    x = pow(a, (t + 1) // 2, Q)
    b = pow(a, t, Q)
    m = s
    while b != 1:
        i, bb = 0, b
        while bb != 1:
            bb = bb * bb % Q
            i += 1
        e = pow(c, 1 << (m - i - 1), Q)
        x = x * e % Q
        c = e * e % Q
        b = b * c % Q
        m = i
    return x


def decompress(b: bytes):
    """JubJubAffine::from_bytes (A.2).  Returns None on failure."""
    assert len(b) == 32
    sign = b[31] >> 7
    bb = bytearray(b)
    bb[31] &= 0x7F
    v = from_le(bytes(bb))
    if v >= Q:
        return None
    v2 = v * v % Q
    den = (1 + D * v2) % Q
    if den == 0:
        return None
    u2 = (v2 - 1) * fq_inv(den) % Q
    u = fq_sqrt(u2)
    if u is None:
        return None
    if (u & 1) != sign:
        if u == 0:
            return None  # u = 0 with the sign bit set: unpinned by the reference's tests; rejected (ZIP 216)
        u = (-u) % Q
    return (u, v)


# ----------------------------------------------------------------------------
# A.3 Poseidon / Hades width 5 over the SAFE sponge
# ----------------------------------------------------------------------------
WIDTH = 5
FULL_ROUNDS = 8
PARTIAL_ROUNDS = 60
N_ROUNDS = FULL_ROUNDS + PARTIAL_ROUNDS


def _gen_round_constants():
    out = []
    h = b"poseidon-for-plonk"
    c = 1
    for _ in range(WIDTH * N_ROUNDS):
        h = hashlib.sha512(h).digest()
        c = (from_le(h) + c) % Q
        out.append(c)
    return out


RC_PLAIN = _gen_round_constants()
# constants as they act on canonical values carry a factor 2^256 (A.3 item 5)
RC = [c * MONT_R % Q for c in RC_PLAIN]
MDS = [[MONT_R * fq_inv(i + j + 5) % Q for j in range(WIDTH)] for i in range(WIDTH)]


def hades_permute(state):
    s = list(state)
    half = FULL_ROUNDS // 2
    for rnd in range(N_ROUNDS):
        for i in range(WIDTH):
            s[i] = (s[i] + RC[WIDTH * rnd + i]) % Q
        if rnd < half or rnd >= half + PARTIAL_ROUNDS:
            s = [pow(x, 5, Q) for x in s]
        else:
            s[4] = pow(s[4], 5, Q)
        s = [sum(MDS[i][j] * s[j] for j in range(WIDTH)) % Q for i in range(WIDTH)]
    return s


def sponge_tag(n_inputs: int, n_outputs: int = 1, domain: int = 0) -> int:
    data = (
        (0x80000000 | n_inputs).to_bytes(4, "big")
        + n_outputs.to_bytes(4, "big")
        + domain.to_bytes(8, "big")
    )
    return from_le(hashlib.blake2b(data, digest_size=64).digest()) % Q


def poseidon_digest(inputs) -> int:
    """Hash::digest(Domain::Other, inputs)[0], untruncated."""
    state = [sponge_tag(len(inputs)), 0, 0, 0, 0]
    pos = 0
    for x in inputs:
        if pos == 4:
            state = hades_permute(state)
            pos = 0
        state[1 + pos] = (state[1 + pos] + x) % Q
        pos += 1
    state = hades_permute(state)
    return state[1]


def digest_truncated(inputs) -> int:
    """Hash::digest_truncated(Domain::Other, inputs)[0] -> JubJubScalar (250 bits)."""
    return poseidon_digest(inputs) & ((1 << 250) - 1)


# ----------------------------------------------------------------------------
# A.4 transcripts
# ----------------------------------------------------------------------------
DOUBLE_CHALLENGE_DOMAIN = int.from_bytes(b"JJSCHDBL", "big")
assert DOUBLE_CHALLENGE_DOMAIN == 0x4A4A53434844424C


def challenge_single(R, PK, m: int) -> int:
    """src/signatures.rs:122-140"""
    return digest_truncated([R[0], R[1], PK[0], PK[1], m])


def challenge_double(R, Rp, PK, PKp, m: int) -> int:
    """src/signatures/double.rs:151-177"""
    return digest_truncated(
        [DOUBLE_CHALLENGE_DOMAIN, R[0], R[1], Rp[0], Rp[1], PK[0], PK[1], PKp[0], PKp[1], m]
    )


def challenge_vargen(R, PK, Gen, m: int) -> int:
    """src/signatures/var_gen.rs:121-142"""
    return digest_truncated([R[0], R[1], PK[0], PK[1], Gen[0], Gen[1], m])


# ----------------------------------------------------------------------------
# verify (the hot path)
# ----------------------------------------------------------------------------
def _canonical(points, fqs, scalars) -> bool:
    for p in points:
        if p[0] >= Q or p[1] >= Q:
            return False
    for x in fqs:
        if x >= Q:
            return False
    for s in scalars:
        if s >= R_ORDER:
            return False
    return True


def _equation(base, u: int, PK, c: int, R) -> bool:
    """u*base + c*PK == R, projective comparison (src/keys/public.rs:128-130)."""
    lhs = _ext_add(mul_ext(base, u), mul_ext(PK, c))
    x, y, z, _ = lhs
    return (x - R[0] * z) % Q == 0 and (y - R[1] * z) % Q == 0


def verify_single(u: int, R, PK, m: int) -> int:
    """PublicKey::verify (src/keys/public.rs:114-135)."""
    if not _canonical([R, PK], [m], [u]):
        return MALFORMED
    if not point_is_valid(PK) or not point_is_valid(R):
        return INVALID_POINT
    c = challenge_single(R, PK, m)
    return OK if _equation(G, u, PK, c, R) else INVALID_SIGNATURE


def verify_double(u: int, R, Rp, PK, PKp, m: int) -> int:
    """PublicKeyDouble::verify (src/keys/public/double.rs:86-117)."""
    if not _canonical([R, Rp, PK, PKp], [m], [u]):
        return MALFORMED
    if not (point_is_valid(PK) and point_is_valid(PKp) and point_is_valid(R) and point_is_valid(Rp)):
        return INVALID_POINT
    c = challenge_double(R, Rp, PK, PKp, m)
    ok = _equation(G, u, PK, c, R) and _equation(G_NUMS, u, PKp, c, Rp)
    return OK if ok else INVALID_SIGNATURE


def verify_vargen(u: int, R, PK, Gen, m: int) -> int:
    """PublicKeyVarGen::verify (src/keys/public/var_gen.rs:107-133)."""
    if not _canonical([R, PK, Gen], [m], [u]):
        return MALFORMED
    if not (point_is_valid(PK) and point_is_valid(Gen) and point_is_valid(R)):
        return INVALID_POINT
    c = challenge_vargen(R, PK, Gen, m)
    return OK if _equation(Gen, u, PK, c, R) else INVALID_SIGNATURE


# ----------------------------------------------------------------------------
# signing: producer of test inputs only
# ----------------------------------------------------------------------------
TAG_STANDARD, TAG_DOUBLE = 1, 2


def sign_single_with_rand(sk: int, rand: int, m: int):
    """SecretKey::sign with the RNG draw made explicit (src/keys/secret.rs:174-194,
    src/nonce.rs:32-44).  Returns (u, R_affine)."""
    r = digest_truncated([rand, sk, TAG_STANDARD, m])
    R = mul(G, r)
    PK = mul(G, sk)
    c = challenge_single(R, PK, m)
    return (r - c * sk) % R_ORDER, R


def sign_double_with_rand(sk: int, rand: int, m: int):
    """SecretKey::sign_double (src/keys/secret/double.rs:56-85, src/nonce.rs:49-61)."""
    r = digest_truncated([rand, sk, TAG_DOUBLE, m])
    R, Rp = mul(G, r), mul(G_NUMS, r)
    PK, PKp = mul(G, sk), mul(G_NUMS, sk)
    c = challenge_double(R, Rp, PK, PKp, m)
    return (r - c * sk) % R_ORDER, R, Rp


def sign_vargen_with_rand(sk: int, Gen, rand: int, m: int):
    """SecretKeyVarGen::sign (src/keys/secret/var_gen.rs:228-256, src/nonce.rs:68-85)."""
    r = digest_truncated([rand, sk, Gen[0], Gen[1], m])
    R = mul(Gen, r)
    PK = mul(Gen, sk)
    c = challenge_vargen(R, PK, Gen, m)
    return (r - c * sk) % R_ORDER, R


# ----------------------------------------------------------------------------
# A.5 StdRng::seed_from_u64 clone (ChaCha12) -- reproduces the reference's seeded tests
# ----------------------------------------------------------------------------
class StdRng:
    def __init__(self, seed_u64: int):
        state = seed_u64 & 0xFFFFFFFFFFFFFFFF
        key = []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & 0xFFFFFFFFFFFFFFFF
            xs = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
            rot = state >> 59
            key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF)
        self.key = key
        self.counter = 0
        self.buf = b""

    @staticmethod
    def _qr(s, a, b, c, d):
        M = 0xFFFFFFFF
        s[a] = (s[a] + s[b]) & M; s[d] ^= s[a]; s[d] = ((s[d] << 16) | (s[d] >> 16)) & M
        s[c] = (s[c] + s[d]) & M; s[b] ^= s[c]; s[b] = ((s[b] << 12) | (s[b] >> 20)) & M
        s[a] = (s[a] + s[b]) & M; s[d] ^= s[a]; s[d] = ((s[d] << 8) | (s[d] >> 24)) & M
        s[c] = (s[c] + s[d]) & M; s[b] ^= s[c]; s[b] = ((s[b] << 7) | (s[b] >> 25)) & M

    def _block(self):
        init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + self.key + [
            self.counter & 0xFFFFFFFF, (self.counter >> 32) & 0xFFFFFFFF, 0, 0]
        s = list(init)
        for _ in range(6):
            self._qr(s, 0, 4, 8, 12); self._qr(s, 1, 5, 9, 13)
            self._qr(s, 2, 6, 10, 14); self._qr(s, 3, 7, 11, 15)
            self._qr(s, 0, 5, 10, 15); self._qr(s, 1, 6, 11, 12)
            self._qr(s, 2, 7, 8, 13); self._qr(s, 3, 4, 9, 14)
        self.counter += 1
        return b"".join(((s[i] + init[i]) & 0xFFFFFFFF).to_bytes(4, "little") for i in range(16))

    def fill_bytes(self, n: int) -> bytes:
        while len(self.buf) < n:
            self.buf += self._block()
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def random_fr(self) -> int:
        """JubJubScalar::random = from_bytes_wide(64 bytes)."""
        return from_le(self.fill_bytes(64)) % R_ORDER

    def random_fq(self) -> int:
        """BlsScalar::random = from_bytes_wide(64 bytes)."""
        return from_le(self.fill_bytes(64)) % Q


def sign_single(rng: StdRng, sk: int, m: int):
    return sign_single_with_rand(sk, rng.random_fr(), m)


def sign_double(rng: StdRng, sk: int, m: int):
    return sign_double_with_rand(sk, rng.random_fr(), m)


def sign_vargen(rng: StdRng, sk: int, Gen, m: int):
    return sign_vargen_with_rand(sk, Gen, rng.random_fr(), m)


# ----------------------------------------------------------------------------
# multisig transcript -- only to consume the reference KAT (src/multisig.rs:393-500)
# ----------------------------------------------------------------------------
def multisig_transcript(pks, Rs, Ss, m: int):
    ds = []
    agg = IDENTITY
    for pk in pks:
        pre = [pk[0], pk[1]]
        for p in pks:
            pre += [p[0], p[1]]
        d = digest_truncated(pre)
        ds.append(d)
        agg = add(agg, mul(pk, d))
    pre = [agg[0], agg[1], m]
    for Rp, Sp in zip(Rs, Ss):
        pre += [Rp[0], Rp[1], Sp[0], Sp[1]]
    a = digest_truncated(pre)
    rsa = IDENTITY
    for Rp, Sp in zip(Rs, Ss):
        rsa = add(add(rsa, Rp), mul(Sp, a))
    c = digest_truncated([rsa[0], rsa[1], agg[0], agg[1], m])
    return ds, agg, a, rsa, c


def multisig_sign_share(sk: int, r: int, s: int, pks, Rs, Ss, m: int) -> int:
    """sign_round_2 (src/multisig.rs:213-257) without the structural checks: z = r + s*a - c*d_i*sk."""
    ds, _, a, _, c = multisig_transcript(pks, Rs, Ss, m)
    i = pks.index(mul(G, sk))
    return (r + s * a - c * ds[i] * sk) % R_ORDER


def multisig_verify_share(z: int, index: int, pks, Rs, Ss, m: int) -> bool:
    """verify_share (src/multisig.rs:284-309, 366-387): z*G + (c*d_i)*PK_i == R_i + a*S_i."""
    ds, _, a, _, c = multisig_transcript(pks, Rs, Ss, m)
    lhs = add(mul(G, z), mul(pks[index], c * ds[index] % R_ORDER))
    rhs = add(Rs[index], mul(Ss[index], a))
    return lhs == rhs


def multisig_combine(zs, pks, Rs, Ss, m: int):
    """combine (src/multisig.rs:326-360): (None, first bad index) or ((u, RSa), None)."""
    ds, _, a, rsa, c = multisig_transcript(pks, Rs, Ss, m)
    for i, z in enumerate(zs):
        lhs = add(mul(G, z), mul(pks[i], c * ds[i] % R_ORDER))
        if lhs != add(Rs[i], mul(Ss[i], a)):
            return None, i
    return (sum(zs) % R_ORDER, rsa), None


# ----------------------------------------------------------------------------
# base58 (bitcoin alphabet), for tests/serde.rs vectors
# ----------------------------------------------------------------------------
_B58 = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"


def b58decode(s: str) -> bytes:
    n = 0
    for ch in s:
        n = n * 58 + _B58.index(ch)
    pad = len(s) - len(s.lstrip("1"))
    body = n.to_bytes((n.bit_length() + 7) // 8, "big") if n else b""
    return b"\x00" * pad + body


# small-order helpers for negative tests
ORDER2 = (0, Q - 1)
