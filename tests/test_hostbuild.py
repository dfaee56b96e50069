"""The product's arithmetic and verification source (jubjub_schnorr_amd/csrc/*.h) compiled for the
CPU and checked against the oracle.  The GPU tests (-m gpu) check the same source as HIP kernels."""
import numpy as np
import pytest

import hostlib as hl
import jjs_oracle as o
import jjs_oracle_c as oc
from helpers import (edge_cases, fe_arr, fe_bytes, make_batch, oracle_verify, pt_arr, rand_mod, to_int, torsion_generator,
                     torsion_grid)


def special_fq(rng, n):
    a = rand_mod(rng, n, o.Q)
    specials = [0, 1, 2, o.Q - 1, o.Q - 2, (1 << 255) % o.Q, (1 << 261) % o.Q, (1 << 29) - 1, (1 << 232), (o.Q + 1) // 2]
    for i, s in enumerate(specials):
        a[i] = fe_bytes(s % o.Q)
    return a


def test_field_ops_match_bigint():
    rng = np.random.default_rng(3)
    a, b = special_fq(rng, 256), special_fq(rng, 256)[::-1].copy()
    prod = hl.fq_mul(a, b); sq = hl.fq_sqr(a); d, s = hl.fq_addsub(a, b)
    for i in range(len(a)):
        x, y = to_int(a[i]), to_int(b[i])
        assert to_int(prod[i]) == x * y % o.Q
        assert to_int(sq[i]) == x * x % o.Q
        assert to_int(d[i]) == (x - y) % o.Q
        assert to_int(s[i]) == (x + y) % o.Q
    inv = hl.fq_inv(a[:16])
    for i in range(16):
        x = to_int(a[i])
        assert to_int(inv[i]) == (pow(x, o.Q - 2, o.Q) if x else 0)


def test_fq_inverse_by_division_steps():
    """csrc/fq_inv.h (safegcd division steps, 20 batches of 30) against a^(q-2): the values at which such code breaks -- 0, 1, 2,
    q - 1, q - 2, (q + 1) / 2, powers of two around the limb boundaries of both radices, all-ones patterns -- and 3 000 random ones."""
    rng = np.random.default_rng(77)
    vals = [0, 1, 2, 3, o.Q - 1, o.Q - 2, (o.Q + 1) // 2, (o.Q - 1) // 2, 2**254, 2**254 + 1, 2**255 - 19 - o.Q if 2**255 - 19 > o.Q else 5]
    vals += [2**k for k in (29, 30, 31, 32, 58, 59, 60, 64, 87, 90, 120, 232, 240, 253)] + [2**k - 1 for k in (29, 30, 60, 90, 240, 254)]
    vals += [o.Q - 2**k for k in (1, 29, 30, 60, 200)] + [pow(3, k, o.Q) for k in (100, 1000, 10**6)]
    vals = [v % o.Q for v in vals]
    a = np.concatenate([np.stack([np.frombuffer(int(v).to_bytes(32, "little"), np.uint8) for v in vals]), rand_mod(rng, 3000, o.Q)])
    inv = hl.fq_inv(a)
    for i in range(len(a)):
        x = to_int(a[i])
        assert to_int(inv[i]) == (pow(x, o.Q - 2, o.Q) if x else 0), hex(x)


def test_poseidon_matches_oracle():
    rng = np.random.default_rng(4)
    for k in (1, 4, 5, 7, 8, 10, 15):
        x = rand_mod(rng, 4 * k, o.Q).reshape(4, k, 32)
        x[0, :, :] = fe_bytes(o.Q - 1)
        assert (hl.poseidon(x) == oc.poseidon(x)).all()


def test_point_flags_all_cosets():
    t8 = torsion_generator()
    s = o.mul(o.G, 987654321)
    rng = np.random.default_rng(8)
    pts = [o.mul(t8, k) for k in range(8)]
    for _ in range(12):
        s = o.mul(o.G, int.from_bytes(rng.bytes(31), "little") + 1)
        pts += [o.add(s, o.mul(t8, k)) for k in range(8)]
    got, want = hl.point_flags(pt_arr(pts)), oc.point_flags(pt_arr(pts))
    assert ((got & 7) == want).all()            # pairing test == the reference's predicate
    assert (((got >> 3) & 1) == ((want >> 1) & 1)).all()   # and so is the [r]P cross-check


def test_comb_tables_are_multiples_of_generators():
    bits = hl.comb_bits()
    top, last = (1 << bits) - 1, 256 // bits - 1
    for which, base in ((0, o.G), (1, o.G_NUMS)):
        for i, b in ((0, 0), (0, 1), (0, top), (1, 1), (last // 2, 200), (last // 2, top - 5), (last, 1), (last, 15)):
            p = o.mul(base, b << (bits * i)) if b else o.IDENTITY
            e = hl.comb_entry(which, i, b)
            ypx, ymx, t2d = (to_int(e[32 * k:32 * k + 32]) for k in range(3))
            assert ypx == (p[1] + p[0]) % o.Q and ymx == (p[1] - p[0]) % o.Q
            assert t2d == 2 * o.D * p[0] * p[1] % o.Q
            # the per-entry builder the device runs gives the same entry as the row filler of the CPU harness
            assert hl.comb_entry_matches_device_builder(which, i, b)


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_verify_matches_oracle_on_mixed_batch(scheme):
    b = make_batch(scheme, 80, seed=21, n_keys=8)
    want, want_c = oracle_verify(scheme, b, want_c=True)
    st, tally, c = hl.verify(scheme, b, want_c=True)
    assert st.tolist() == want.tolist()
    assert (c == want_c).all()
    assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
    assert len(set(want.tolist())) >= 3


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_verify_matches_oracle_on_edge_cases(scheme):
    b = edge_cases(scheme)
    want = oracle_verify(scheme, b)
    st, _ = hl.verify(scheme, b)
    assert st.tolist() == want.tolist()


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_every_torsion_component_is_caught(scheme):
    """The first pass tests combinations of points (one pairing test per equation); the statuses must still be
    the per-point ones of the reference for every choice of small-order components."""
    b = torsion_grid(scheme, extra=0 if scheme == "single" else 150)
    want = oracle_verify(scheme, b)
    st, tally = hl.verify(scheme, b)
    assert st.tolist() == want.tolist()
    assert set(want.tolist()) == {0, 1, 2}
    assert tally.tolist() == [int((want == k).sum()) for k in range(4)]


def half_size_cases():
    """Challenges for the truncated Euclid: random ones plus inputs that force huge partial quotients, remainders
    next to 2^126 and the uncertified-step fallback of the Lehmer loop."""
    rng = np.random.default_rng(12)
    c = rand_mod(rng, 3000, 1 << 250)
    R = o.R_ORDER
    specials = [0, 1, 2, (1 << 126) - 1, 1 << 126, (1 << 126) + 1, (1 << 250) - 1, R - 1, R // 2,
                (R + 1) // 2, 1 << 127, 1 << 200, 3 << 248, R - (1 << 126), (1 << 125) + 12345]
    # c = floor(r / k) and neighbours: the first quotient is k (up to 2^120: far beyond the 2^26 a Lehmer run certifies)
    for k in (3, 1 << 26, (1 << 26) - 1, (1 << 26) + 1, 1 << 31, (1 << 31) + 1, 1 << 32, 1 << 63, 1 << 120, 1 << 125):
        specials += [R // k, R // k + 1]
    # continued fractions with a run of ones (slowest convergence), then one huge quotient
    a, b = 1, 1
    for _ in range(120):
        a, b = a + b, a
    specials += [R * b // a, (R * b // a) | 1, (R >> 124) << 123]
    # remainders that land within a few units of 2^126 after one step: c = (r - (2^126 + d)) / q for small q
    for d in (-2, -1, 0, 1, 2):
        for q in (1, 2, 5):
            specials.append((R - ((1 << 126) + d)) // q)
    specials = [x % R for x in specials]
    for i, x in enumerate(specials):
        c[i] = fe_bytes(x)
    return c, len(specials)


def check_half_size(c, a_bytes, b_bytes, neg):
    """a = b*c (mod r), a, |b| < 2^126, b != 0, and (a, b) is exactly where the textbook Euclid on (r, c) first
    drops below 2^126: every step the implementation takes (single- or multi-quotient, on truncated operands) must
    be a true Euclid step."""
    for i in range(len(c)):
        ci = to_int(c[i])
        a = int.from_bytes(bytes(a_bytes[i]), "little")
        b = int.from_bytes(bytes(b_bytes[i]), "little")
        if neg[i]:
            b = -b
        assert 0 <= a < 1 << 126 and 0 < abs(b) < 1 << 126, (i, a, b)
        assert (a - b * ci) % o.R_ORDER == 0, i
        r0, r1, t0, t1 = o.R_ORDER, ci, 0, 1
        while r1 >= 1 << 126:
            q = r0 // r1
            r0, r1, t0, t1 = r1, r0 - q * r1, t1, t0 - q * t1
        assert (a, b) == (r1, t1), i


def test_half_size_scalars():
    """The CPU build of half_size_scalars (1.0 / y estimates, per-lane loop control); the device code path (v_rcp_f64,
    wave ballots) runs the same cases in tests/test_gpu_parity.py::test_half_size_scalars_on_device."""
    c, n_special = half_size_cases()
    assert n_special >= 15 and len(c) >= 3000
    out = hl.half_size(c)
    check_half_size(c, out[:, :16], out[:, 16:32], out[:, 32])


def wire_point_cases(rng, n_random=64):
    """Compressed encodings covering every decode branch, with the oracle's answer."""
    enc = []
    for _ in range(n_random):
        p = o.mul(o.G, int.from_bytes(rng.bytes(31), "little") + 1)
        enc.append(o.compress(p))
    t8 = torsion_generator()
    enc += [o.compress(o.mul(t8, k)) for k in range(8)]              # small-order points decode fine
    enc += [o.compress(o.IDENTITY), o.compress(o.ORDER2)]
    ident_bad = bytearray(o.compress(o.IDENTITY)); ident_bad[31] |= 0x80   # u = 0, sign bit set
    o2_bad = bytearray(o.compress(o.ORDER2)); o2_bad[31] |= 0x80
    enc += [bytes(ident_bad), bytes(o2_bad)]
    enc.append(o.le32(o.Q))                                              # v = q (non canonical)
    enc.append(bytes([0xFF] * 31 + [0x7F]))                              # v = 2^255 - 1
    v = 2
    while len(enc) < n_random + 24:                                      # v with no square root
        b = o.le32(v)
        if o.decompress(b) is None:
            enc.append(b)
        v += 1
    for i in range(n_random // 2):                                      # sign bit flipped -> the negated point
        b = bytearray(enc[i]); b[31] ^= 0x80; enc.append(bytes(b))
    return enc


def test_decompress_matches_oracle():
    rng = np.random.default_rng(31)
    enc = wire_point_cases(rng)
    arr = np.frombuffer(b"".join(enc), np.uint8).reshape(-1, 32)
    out, ok = hl.decompress(arr)
    n_fail = 0
    for i, e in enumerate(enc):
        want = o.decompress(e)
        assert bool(ok[i]) == (want is not None), i
        if want is None:
            n_fail += 1
            assert out[i].tobytes() == o.le32(0) + o.le32(1)
        else:
            assert out[i].tobytes() == o.le32(want[0]) + o.le32(want[1]), i
    assert n_fail >= 10


def test_safe_tag_computed_at_call_time():
    """csrc/safe_tag.h (BLAKE2b and the reduction, written for the host side of the multisig entry point) against the
    oracle's tag (hashlib) for lengths inside and far outside the generated table, and against the table itself."""
    R261 = pow(2, 261, o.Q)
    for n in (1, 2, 5, 16, 17, 516, 1027, 1028, 2002, 4003, 100003, (1 << 24) * 4 + 3):
        want = o.sponge_tag(n) * R261 % o.Q
        limbs = [(want >> (29 * i)) & 0x1FFFFFFF for i in range(9)]
        assert hl.safe_tag(n).tolist() == limbs, n
        if n < 1028:
            assert hl.safe_tag(n, table=True).tolist() == limbs, n


def long_multisig_transcript(n: int, seed: int, corrupt=()):
    """One valid transcript of n participants with everything `combine` returns, built with the secret keys so that it
    costs a handful of scalar multiplications instead of 4 n: d_i from the C oracle's sponge (any length), then
    pk_agg = (sum d_i sk_i) G, RSa = (sum r_i + a sum s_i) G, c, z_i = r_i + s_i a - c d_i sk_i (src/multisig.rs:213-257,
    440-500).  Points as the C oracle computes them (sk G etc.).  Returns the arrays and (agg, u, RSa, statuses)."""
    import jjs_oracle_c as oc
    from helpers import pt_bytes, to_int, to_pt
    rng = np.random.default_rng(seed)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (o.R_ORDER - 1) + 1  # noqa: E731
    sks, rs, ss = [rnd() for _ in range(n)], [rnd() for _ in range(n)], [rnd() for _ in range(n)]
    G = np.tile(pt_bytes(o.G), (n, 1))
    PK, R, S = (oc.scalar_mul(G, fe_arr(v)) for v in (sks, rs, ss))
    msg = int.from_bytes(rng.bytes(40), "little") % o.Q
    pre = np.empty((n, 2 + 2 * n, 32), np.uint8)
    pre[:, 0] = PK[:, :32]; pre[:, 1] = PK[:, 32:]
    pre[:, 2::2] = PK[:, :32][None]; pre[:, 3::2] = PK[:, 32:][None]
    ds = [to_int(r) & ((1 << 250) - 1) for r in oc.poseidon_any(pre)]
    agg = o.mul(o.G, sum(d * k for d, k in zip(ds, sks)) % o.R_ORDER)
    pre = np.empty((1, 3 + 4 * n, 32), np.uint8)
    pre[0, 0], pre[0, 1], pre[0, 2] = fe_bytes(agg[0]), fe_bytes(agg[1]), fe_bytes(msg)
    pre[0, 3::4] = R[:, :32]; pre[0, 4::4] = R[:, 32:]; pre[0, 5::4] = S[:, :32]; pre[0, 6::4] = S[:, 32:]
    a = to_int(oc.poseidon_any(pre)[0]) & ((1 << 250) - 1)
    rsa = o.mul(o.G, (sum(rs) + a * sum(ss)) % o.R_ORDER)
    c = o.digest_truncated([rsa[0], rsa[1], agg[0], agg[1], msg])
    zs = [(rs[i] + ss[i] * a - c * ds[i] * sks[i]) % o.R_ORDER for i in range(n)]
    st = [0] * n
    for j in corrupt:
        zs[j] = (zs[j] + 1) % o.R_ORDER; st[j] = 4
    return fe_arr(zs), PK, R, S, fe_arr([msg]), (agg, sum(zs) % o.R_ORDER, rsa, st)


def check_long_multisig(run, sizes=(257, 300)):
    """Transcripts of more than 256 participants (the generated tag table ends there; their tags are computed at call
    time), an empty transcript (InvalidMultisigTranscript for it alone) and ordinary ones in ONE call."""
    from helpers import make_multisig_batch, pt_bytes
    parts, infos, offs = [], [], [0]
    for k, n in enumerate(sizes):
        z, PK, R, S, m, info = long_multisig_transcript(n, seed=40 + k, corrupt=(n - 1,) if k == 1 else ())
        parts.append((z, PK, R, S, m)); infos.append(info); offs.append(offs[-1] + n)
    infos.append(None); offs.append(offs[-1])                  # an empty transcript in the middle
    parts.append((np.empty((0, 32), np.uint8), np.empty((0, 64), np.uint8), np.empty((0, 64), np.uint8), np.empty((0, 64), np.uint8),
                  fe_arr([7])))
    z2, PK2, R2, S2, m2, offs2, want2, info2 = make_multisig_batch(3, seed=77, max_n=4, corrupt=False)
    for t in range(3):
        sl = slice(int(offs2[t]), int(offs2[t + 1]))
        parts.append((z2[sl], PK2[sl], R2[sl], S2[sl], m2[t:t + 1])); infos.append(info2[t] + ([0] * (sl.stop - sl.start),))
        offs.append(offs[-1] + sl.stop - sl.start)
    Z, P_, R_, S_, M = (np.concatenate([p[i] for p in parts]) for i in range(5))
    st, agg, su, sr, ts = run(Z, P_, R_, S_, M, offs)
    for t, info in enumerate(infos):
        lo, hi = offs[t], offs[t + 1]
        if info is None:
            assert ts[t] == 5 and not agg[t].any() and not su[t].any() and not sr[t].any()
            continue
        a_pk, u, rsa, want = info
        assert st[lo:hi].tolist() == list(want), t
        bad = any(want)
        assert ts[t] == (4 if bad else 0), t
        assert agg[t].tobytes() == pt_bytes(a_pk).tobytes(), t
        assert sr[t].tobytes() == (bytes(64) if bad else pt_bytes(rsa).tobytes()), t
        assert su[t].tobytes() == (bytes(32) if bad else o.le32(u)), t


def test_multisig_long_and_empty_transcripts():
    check_long_multisig(hl.multisig, sizes=(257,))


def check_multisig(run, reference_kat):
    from helpers import make_multisig_batch, pt_bytes
    # 1. the reference's KAT transcript (src/multisig.rs:544-672): shares valid, combined signature bytes
    k = reference_kat["multisig_kat"]
    pks = [o.decompress(bytes.fromhex(x)) for x in k["public_keys"]]
    Rs = [o.decompress(bytes.fromhex(x)) for x in k["r_points"]]; Ss = [o.decompress(bytes.fromhex(x)) for x in k["s_points"]]
    z = np.stack([np.frombuffer(bytes.fromhex(x), np.uint8) for x in k["individual_shares"]])
    st, agg, su, sr, ts = run(z, pt_arr(pks), pt_arr(Rs), pt_arr(Ss), fe_arr([k["message"]]), [0, 3])
    assert st.tolist() == [0, 0, 0] and ts.tolist() == [0]
    from helpers import to_pt
    assert o.compress(to_pt(agg[0])).hex() == k["aggregate_public_key"]
    assert (su[0].tobytes() + o.compress(to_pt(sr[0]))).hex() == k["signature"]
    # swapped shares -> InvalidMultisigShare at those slots
    st, _, su_bad, sr_bad, ts = run(z[[1, 0, 2]], pt_arr(pks), pt_arr(Rs), pt_arr(Ss), fe_arr([k["message"]]), [0, 3])
    assert st.tolist() == [4, 4, 0] and ts.tolist() == [4]
    # `combine` returns Err(InvalidMultisigShare), not a signature (src/multisig.rs:340-353): nothing to pick up
    assert not su_bad.any() and not sr_bad.any()
    # a non-canonical message makes every share of the transcript malformed
    st, _, su_bad, sr_bad, ts = run(z, pt_arr(pks), pt_arr(Rs), pt_arr(Ss), fe_arr([o.Q]), [0, 3])
    assert st.tolist() == [3, 3, 3] and ts.tolist() == [3] and not su_bad.any() and not sr_bad.any()
    # 2. random ragged transcripts
    z, PK, R, S, m, offs, want, info = make_multisig_batch(7, seed=9)
    st, agg, su, sr, ts = run(z, PK, R, S, m, offs)
    assert st.tolist() == want.tolist()
    for t, (a_pk, u, rsa) in enumerate(info):
        bad = want[offs[t]:offs[t + 1]].any()
        assert ts[t] == (4 if bad else 0)
        assert agg[t].tobytes() == pt_bytes(a_pk).tobytes()
        assert sr[t].tobytes() == (bytes(64) if bad else pt_bytes(rsa).tobytes())
        assert su[t].tobytes() == (bytes(32) if bad else o.le32(u))


def test_multisig_batch(reference_kat):
    check_multisig(hl.multisig, reference_kat)


# ---- the static bounds of fq29.h, exercised at their edges ------------------------------------------------
RP = 1 << 261


def limbs_val(row):
    return sum(int(x) << (29 * i) for i, x in enumerate(row))


def edge_limb_vectors(rng, n, limb_units, value_units):
    """Limb vectors with every limb < limb_units * 2^29 and value < value_units * q: random, plus vectors
    pushed against both bounds."""
    out = []
    while len(out) < n:
        kind = len(out) % 4
        cap = limb_units << 29
        if kind == 0:
            l = [int(rng.integers(0, cap)) for _ in range(9)]
        elif kind == 1:
            l = [cap - 1] * 9                       # every limb at its maximum
        elif kind == 2:
            l = [cap - 1 - int(rng.integers(0, 4)) for _ in range(9)]
        else:
            l = [int(rng.integers(0, cap)) for _ in range(8)] + [cap - 1]
        # clamp the value below value_units * q by lowering the top limb
        lim = value_units * o.Q - 1
        low = sum(x << (29 * i) for i, x in enumerate(l[:8]))
        l[8] = min(l[8], max(0, (lim - low) >> 232))
        if limbs_val(l) <= lim:
            out.append(l)
    return np.array(out, np.uint64).astype(np.uint32)


def check_product(res, want_mod):
    for i in range(len(res)):
        v = limbs_val(res[i])
        assert all(int(x) < 1 << 29 for x in res[i][:8]), i
        assert v < 2 * o.Q, i
        assert v % o.Q == want_mod[i], i


def test_montgomery_product_at_the_edges_of_its_static_bounds():
    """fq_mul demands La*Lb <= 3 (signed 64-bit columns) and Aa*Ab <= 70; squares take normalised limbs.  The
    result must be exact, in (0, 2q), with normalised limbs."""
    rng = np.random.default_rng(77)
    rinv = pow(RP, -1, o.Q)
    for (la, aa), (lb, ab) in (((1, 2), (1, 2)), ((3, 5), (1, 4)), ((1, 7), (3, 10)), ((1, 35), (3, 2)), ((1, 70), (1, 1)),
                               ((3, 8), (1, 8)), ((1, 1), (3, 70))):
        assert la * lb <= 3 and aa * ab <= 70
        a, b = edge_limb_vectors(rng, 200, la, aa), edge_limb_vectors(rng, 200, lb, ab)
        res = hl.raw_mul(a, b)
        check_product(res, [limbs_val(a[i]) * limbs_val(b[i]) * rinv % o.Q for i in range(len(a))])
        assert all(limbs_val(r) > 0 for r in res)          # the biased last digit keeps the result positive
    zero = np.zeros((1, 9), np.uint32)
    assert limbs_val(hl.raw_mul(zero, zero)[0]) == o.Q      # 0 * 0 comes out as q, never as a negative number
    for la, aa in ((1, 2), (1, 4), (1, 8)):
        assert aa * aa <= 70
        a = edge_limb_vectors(rng, 200, la, aa)
        check_product(hl.raw_sqr(a), [limbs_val(x) ** 2 * rinv % o.Q for x in a])
    assert limbs_val(hl.raw_sqr(zero)[0]) == o.Q


def test_dot_products_at_the_edges_of_their_static_bounds():
    rng = np.random.default_rng(78)
    rinv, inv29 = pow(RP, -1, o.Q), pow(1 << 29, -1, o.Q)
    for row in range(5):
        t = np.stack([edge_limb_vectors(rng, 120, 1, 3) for _ in range(5)], axis=1)      # (n, 5, 9), fe<1,3>
        d, s = hl.raw_dot5(t.reshape(len(t), 45), row)
        mds = [o.MDS[row][j] * RP % o.Q for j in range(5)]                                  # Montgomery constants
        check_product(d, [sum(mds[j] * limbs_val(t[i, j]) for j in range(5)) * rinv % o.Q for i in range(len(t))])
        small = [360360 // (row + j + 5) for j in range(5)]
        check_product(s, [sum(small[j] * limbs_val(t[i, j]) for j in range(5)) * inv29 % o.Q for i in range(len(t))])


def test_golden_vectors_on_the_host_build():
    """tests/golden/verify_vectors.json (the reference's own scenarios) through the CPU build."""
    import json, os
    from helpers import ARG_ORDER
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "verify_vectors.json")))
    H = lambda x: np.frombuffer(bytes.fromhex(x), np.uint8)  # noqa: E731
    for scheme, items in vec.items():
        b = {k: np.stack([H(v[k]) for v in items]) for k in ARG_ORDER[scheme]}
        st, _, c = hl.verify(scheme, b, want_c=True)
        assert st.tolist() == [v["status"] for v in items], scheme
        assert [r.tobytes().hex() for r in c] == [v["c"] for v in items], scheme


# ---- extended-coordinate inputs (csrc/normalize.h) --------------------------------------------------------
def test_normalize_extended_points():
    """(U, V, Z) -> (U/Z, V/Z) with one inversion per lane: any lane count, ragged item counts, 2..4 points per
    item; Z = 0 gives the off-curve marker (0, 0); a coordinate >= q flags the item."""
    from helpers import to_extended
    rng = np.random.default_rng(17)
    for k, n, lanes in ((2, 1, 1), (2, 37, 1), (3, 37, 5), (4, 64, 64), (4, 50, 200), (2, 300, 7)):
        aff = [pt_arr([o.mul(o.G, int.from_bytes(rng.bytes(20), "little") + 1) for _ in range(n)]) for _ in range(k)]
        ext = [to_extended(a, rng) for a in aff]
        zero_rows, bad_rows = set(), set()
        if n > 8:
            ext[0][3, 64:] = 0; zero_rows.add((0, 3))                      # Z = 0
            ext[k - 1][5, 64:] = 0; zero_rows.add((k - 1, 5))
            ext[0][6, 64:] = fe_bytes(o.Q); bad_rows.add(6)                 # Z = q: not canonical
            ext[k - 1][7, :32] = 0xFF; bad_rows.add(7)                      # U = 2^256 - 1
        outs, bad = hl.normalize(ext, lanes)
        assert set(np.nonzero(bad)[0].tolist()) == bad_rows
        for j in range(k):
            for i in range(n):
                if i in bad_rows:
                    continue
                want = bytes(64) if (j, i) in zero_rows else aff[j][i].tobytes()
                assert outs[j][i].tobytes() == want, (k, n, lanes, j, i)


# ---- latency path for small batches (csrc/small_batch.h) -----------------------------------------------------
@pytest.mark.parametrize("positions", [4, 8, 16])
@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_small_batch_path_matches_oracle(scheme, positions):
    """The same statuses from the path that cuts a signature into chain / point / hash / piece lanes: mixed batch,
    hand-built edge cases and every pair of small-order components."""
    b = make_batch(scheme, 48, seed=61, n_keys=8)
    want = oracle_verify(scheme, b)
    st, tally = hl.verify_small(scheme, b, positions)
    assert st.tolist() == want.tolist()
    assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
    b = edge_cases(scheme)
    assert hl.verify_small(scheme, b, positions)[0].tolist() == oracle_verify(scheme, b).tolist()
    b = torsion_grid(scheme, reps=1, extra=0 if scheme == "single" else 40)
    assert hl.verify_small(scheme, b, positions)[0].tolist() == oracle_verify(scheme, b).tolist()


# ---- key-table path (csrc/key_tables.h) -------------------------------------------------------------------
@pytest.mark.parametrize("window", [5, 6])
@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_key_table_path_matches_oracle(scheme, window):
    """Per-key validity and window tables (both widths the engine chooses between), per-item additions only: the same
    statuses as the oracle on a mixed batch with few keys, on the hand-built edge cases (every item its own key) and
    on the torsion grid."""
    b = make_batch(scheme, 60, seed=83, n_keys=4)
    want = oracle_verify(scheme, b)
    st, tally = hl.verify_keyed(scheme, b, window)
    assert st.tolist() == want.tolist()
    assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
    b = edge_cases(scheme)
    assert hl.verify_keyed(scheme, b, window)[0].tolist() == oracle_verify(scheme, b).tolist()
    b = torsion_grid(scheme, reps=1, extra=0 if scheme == "single" else 30)
    assert hl.verify_keyed(scheme, b, window)[0].tolist() == oracle_verify(scheme, b).tolist()


# ---- the first pass in two launches (verify_core.h prep_phase) ---------------------------------------------
@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_prepare_in_two_launches_matches_oracle(scheme):
    """While the keys of a batch are being counted the device runs the first pass as PREP_HEAD (what does not depend on
    the decision) and later PREP_TAIL (what only the throughput path needs), with the record stored in between: the
    same statuses as the oracle on the throughput path, and on the key-table path, which gets the head alone."""
    hl.set_split_prepare(True)
    try:
        for b in (make_batch(scheme, 60, seed=85, n_keys=4), edge_cases(scheme), torsion_grid(scheme, reps=1, extra=0 if scheme == "single" else 30)):
            want = oracle_verify(scheme, b)
            st, tally = hl.verify(scheme, b)
            assert st.tolist() == want.tolist()
            assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
            assert hl.verify_keyed(scheme, b, 6)[0].tolist() == want.tolist()
    finally:
        hl.set_split_prepare(False)
