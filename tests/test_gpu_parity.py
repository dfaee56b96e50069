"""GPU parity: the HIP kernels, called through the C ABI, against the oracle on the same inputs.
Bit-exact everywhere (integer work): status bytes, tallies, challenges, signatures."""
import json
import os

import numpy as np
import pytest

import jjs_oracle as o
import jjs_oracle_c as oc
from helpers import (ARG_ORDER, to_wire, batch_to_extended, edge_cases, ext_on_device, fe_bytes, make_batch, oracle_verify, pt_arr, rand_mod, to_int, torsion_generator,
                     torsion_grid)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import jubjub_schnorr_amd as jjs
    return jjs.engine()


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def test_loaded_library_is_the_in_tree_hip_build(eng):
    import jubjub_schnorr_amd as jjs
    assert os.path.exists(jjs.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert jjs.LIB_PATH in maps


def test_fq_mul(eng):
    rng = np.random.default_rng(1)
    a, b = rand_mod(rng, 4096, o.Q), rand_mod(rng, 4096, o.Q)
    for i, s in enumerate([0, 1, o.Q - 1, o.Q - 2, (1 << 255) % o.Q, (1 << 261) % o.Q, (1 << 29) - 1, 1 << 232]):
        a[i] = fe_bytes(s); b[-1 - i] = fe_bytes(s)
    out = host(eng.debug_fq_mul(dev(a), dev(b)))
    for i in range(len(a)):
        assert to_int(out[i]) == to_int(a[i]) * to_int(b[i]) % o.Q, i


@pytest.mark.parametrize("k", [1, 4, 5, 7, 8, 10, 15, 16])
def test_poseidon(eng, k):
    rng = np.random.default_rng(k)
    x = rand_mod(rng, 300 * k, o.Q).reshape(300, k, 32)
    x[0] = fe_bytes(o.Q - 1); x[1] = fe_bytes(0)
    assert (host(eng.debug_poseidon(dev(x))) == oc.poseidon(x)).all()


def test_point_flags(eng):
    t8 = torsion_generator()
    rng = np.random.default_rng(2)
    pts = []
    for _ in range(24):
        s = o.mul(o.G, int.from_bytes(rng.bytes(31), "little"))
        pts += [o.add(s, o.mul(t8, k)) for k in range(8)]
    pts += [o.mul(t8, k) for k in range(8)] + [(5, 7), (0, 0), (1, 1)]
    arr = pt_arr(pts)
    got, want = host(eng.debug_point_flags(dev(arr))), oc.point_flags(arr)
    # [r]P of an off-curve point depends on the formulas used; only on-curve points have a torsion bit
    off = (want & 1) == 0
    by_order = (got >> 3) & 1
    got &= 7
    got[off] &= 0b101; want[off] &= 0b101
    assert (got == want).all()                              # pairing-based subgroup test
    assert (by_order[~off] == ((want[~off] >> 1) & 1)).all()  # [r]P cross-check
    assert (want[:8] == [3, 1, 1, 1, 1, 1, 1, 1]).all()


def test_half_size_scalars_on_device(eng):
    """The Euclid of the verify kernels on the code path the GPU takes (v_rcp_f64 quotient estimates, wave-ballot
    loop control): random and adversarial challenges, compared with the textbook truncated Euclid.  Also with the
    adversarial items spread so that they share waves with ordinary ones (the loop runs until the slowest lane)."""
    from test_hostbuild import check_half_size, half_size_cases
    c, n_special = half_size_cases()
    a, b, neg = (host(t) for t in eng.debug_half_scalars(dev(c)))
    check_half_size(c, a, b, neg)
    perm = np.random.default_rng(3).permutation(len(c))
    a2, b2, neg2 = (host(t) for t in eng.debug_half_scalars(dev(c[perm])))
    assert (a2 == a[perm]).all() and (b2 == b[perm]).all() and (neg2 == neg[perm]).all()
    # ragged sizes: a one-lane wave and a partly filled last wave
    for n in (1, 65):
        a3, b3, neg3 = (host(t) for t in eng.debug_half_scalars(dev(c[:n])))
        assert (a3 == a[:n]).all() and (b3 == b[:n]).all() and (neg3 == neg[:n]).all()


def test_comb_tables(eng):
    RPI = pow(1 << 261, -1, o.Q)
    for which, base in ((0, o.G), (1, o.G_NUMS)):
        tab = eng.debug_comb_table(which)
        windows, entries = tab.shape[0], tab.shape[1]
        bits = entries.bit_length() - 1
        for i, b in ((0, 0), (0, 1), (0, entries - 1), (1, 1), (5, 77), (windows // 2, entries // 2), (windows - 1, 1),
                     (windows - 1, 15), (windows - 1, entries - 3)):
            p = o.mul(base, b << (bits * i)) if b else o.IDENTITY
            e = tab[i, b]
            val = [sum(int(x) << (29 * j) for j, x in enumerate(e[9 * c:9 * c + 9])) * RPI % o.Q for c in range(3)]
            assert val == [(p[1] + p[0]) % o.Q, (p[1] - p[0]) % o.Q, 2 * o.D * p[0] * p[1] % o.Q]


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
@pytest.mark.parametrize("n", [1, 63, 65, 1000])
def test_verify_mixed_batch_dev(eng, scheme, n):
    b = make_batch(scheme, n, seed=100 + n, n_keys=16)
    want, want_c = oracle_verify(scheme, b, want_c=True)
    args = [dev(b[k]) for k in ARG_ORDER[scheme]]
    st, tally = eng.verify(scheme, *args)
    c = eng.challenge(scheme, *args[1:])
    assert host(st).tolist() == want.tolist()
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]
    assert (host(c) == want_c).all()


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_verify_host_buffers_and_edge_cases(eng, scheme):
    b = edge_cases(scheme)
    want = oracle_verify(scheme, b)
    st, tally = eng.verify(scheme, *[b[k] for k in ARG_ORDER[scheme]])
    assert st.tolist() == want.tolist()
    assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
    assert set(want.tolist()) == {0, 1, 2, 3}


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_every_torsion_component_is_caught(eng, scheme):
    """Small-order components on every point, with the prime-order parts satisfying the equation or not: the
    two-pass subgroup logic (combined pairing test, resolve pass) must give the reference's per-point statuses."""
    b = torsion_grid(scheme, extra=0 if scheme == "single" else 300)
    want = oracle_verify(scheme, b)
    st, tally = eng.verify(scheme, *[dev(b[k]) for k in ARG_ORDER[scheme]])
    assert host(st).tolist() == want.tolist()
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]
    _, tally_only = eng.verify(scheme, *[dev(b[k]) for k in ARG_ORDER[scheme]], want_status=False)
    assert host(tally_only).tolist() == host(tally).tolist()


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_empty_batch(eng, scheme):
    import torch
    widths = {"single": (32, 64, 64, 32), "double": (32, 64, 64, 64, 64, 32), "vargen": (32, 64, 64, 64, 32)}[scheme]
    st, tally = eng.verify(scheme, *[np.zeros((0, w), np.uint8) for w in widths])
    assert len(st) == 0 and tally.tolist() == [0, 0, 0, 0]
    st, tally = eng.verify(scheme, *[torch.zeros((0, w), dtype=torch.uint8, device="cuda") for w in widths])
    assert st.numel() == 0 and host(tally).tolist() == [0, 0, 0, 0]


def test_golden_vectors(eng):
    vec = json.load(open(os.path.join(GOLDEN, "verify_vectors.json")))
    H = lambda x: np.frombuffer(bytes.fromhex(x), np.uint8)  # noqa: E731
    for scheme, items in vec.items():
        arrays = [np.stack([H(v[k]) for v in items]) for k in ARG_ORDER[scheme]]
        st, _ = eng.verify(scheme, *arrays)
        assert st.tolist() == [v["status"] for v in items], scheme
        c = host(eng.challenge(scheme, *[dev(a) for a in arrays[1:]]))
        assert [row.tobytes().hex() for row in c] == [v["c"] for v in items]


def test_reference_shaped_api(eng):
    import jubjub_schnorr_amd as jjs
    vec = json.load(open(os.path.join(GOLDEN, "verify_vectors.json")))
    B = bytes.fromhex
    v = vec["single"][0]
    jjs.PublicKey(B(v["PK"])).verify(jjs.Signature(B(v["u"]), B(v["R"])), B(v["m"]))
    by_name = {x["name"]: x for x in vec["single"]}
    w = by_name["serde_signature_wrong_key"]
    with pytest.raises(jjs.InvalidSignature):
        jjs.PublicKey(B(w["PK"])).verify(jjs.Signature(B(w["u"]), B(w["R"])), B(w["m"]))
    w = by_name["serde_signature_identity_pk"]
    with pytest.raises(jjs.InvalidPoint):
        jjs.PublicKey(B(w["PK"])).verify(jjs.Signature(B(w["u"]), B(w["R"])), B(w["m"]))
    d = vec["double"][0]
    jjs.PublicKeyDouble(B(d["PK"]), B(d["PKp"])).verify(jjs.SignatureDouble(B(d["u"]), B(d["R"]), B(d["Rp"])), B(d["m"]))
    d = {x["name"]: x for x in vec["double"]}["legacy_double_attack"]
    with pytest.raises(jjs.InvalidSignature):
        jjs.PublicKeyDouble(B(d["PK"]), B(d["PKp"])).verify(jjs.SignatureDouble(B(d["u"]), B(d["R"]), B(d["Rp"])), B(d["m"]))
    g = vec["vargen"][0]
    jjs.PublicKeyVarGen(B(g["PK"]), B(g["Gen"])).verify(jjs.SignatureVarGen(B(g["u"]), B(g["R"])), B(g["m"]))
    assert jjs.PublicKey.verify_batch([]).shape == (0,)


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_gpu_signer_matches_oracle(eng, scheme):
    rng = np.random.default_rng(9)
    n = 200
    sk, rnd, m, g = rand_mod(rng, n, o.R_ORDER, nonzero=True), rand_mod(rng, n, o.R_ORDER), rand_mod(rng, n, o.Q), rand_mod(rng, n, o.R_ORDER, nonzero=True)
    sk[0] = fe_bytes(1); sk[1] = fe_bytes(o.R_ORDER - 1); rnd[2] = fe_bytes(0); m[3] = fe_bytes(0)
    if scheme == "vargen":
        got = eng.sign(scheme, dev(sk), dev(rnd), dev(m), gen_scalar=dev(g))
        want = oc.sign_vargen(sk, g, rnd, m)
    else:
        got = eng.sign(scheme, dev(sk), dev(rnd), dev(m))
        want = (oc.sign_single if scheme == "single" else oc.sign_double)(sk, rnd, m)
    for a, b in zip(got, want):
        assert (host(a) == b).all()


@pytest.mark.parametrize("scheme,log2n", [("single", 20), ("double", 20), ("vargen", 20), ("single", 21)])
def test_full_size_properties(eng, scheme, log2n):
    """BASELINE-size batch: inputs from the GPU signer, known corruption pattern, so the expected
    status of every item is known by construction; plus an oracle check of a random sample.  2^21 single signatures
    are one GPU's shard of BASELINE.json configs[3] (2^24 over 8 GPUs)."""
    import torch
    n = 1 << log2n
    gen = torch.Generator(device="cpu").manual_seed(0x6A6A73 + log2n)
    def scal(mod_top):
        t = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=gen)
        t[:, 31] &= mod_top
        return t.cuda()
    sk, rnd, m, g = scal(0x07), scal(0x07), scal(0x3F), scal(0x07)   # < 2^251 < r, < 2^254 < q
    sk[:, 0] |= 1
    if scheme == "vargen":
        u, R, PK, Gen = eng.sign(scheme, sk, rnd, m, gen_scalar=g)
        arrs = {"u": u, "R": R, "PK": PK, "Gen": Gen, "m": m}
    elif scheme == "double":
        u, R, Rp, PK, PKp = eng.sign(scheme, sk, rnd, m)
        arrs = {"u": u, "R": R, "Rp": Rp, "PK": PK, "PKp": PKp, "m": m}
    else:
        u, R, PK = eng.sign(scheme, sk, rnd, m)
        arrs = {"u": u, "R": R, "PK": PK, "m": m}
    idx = torch.arange(n, device="cuda")
    expect = torch.zeros(n, dtype=torch.uint8, device="cuda")
    bad_sig = (idx % 16) == 3          # tampered message -> InvalidSignature
    arrs["m"] = arrs["m"].clone(); arrs["m"][bad_sig, 0] ^= 1
    expect[bad_sig] = 2
    bad_pt = (idx % 64) == 7           # identity public key -> InvalidPoint (wins over InvalidSignature)
    ident = torch.zeros(64, dtype=torch.uint8, device="cuda"); ident[32] = 1
    arrs["PK"] = arrs["PK"].clone(); arrs["PK"][bad_pt] = ident
    expect[bad_pt] = 1
    malformed = (idx % 1024) == 11     # u = 2^256 - 1 -> Malformed
    arrs["u"] = arrs["u"].clone(); arrs["u"][malformed] = 0xFF
    expect[malformed] = 3
    st, tally = eng.verify(scheme, *[arrs[k].contiguous() for k in ARG_ORDER[scheme]])
    assert torch.equal(st, expect)
    assert host(tally).tolist() == [int((expect == k).sum()) for k in range(4)]
    # idempotence: same inputs, same outputs
    st2, tally2 = eng.verify(scheme, *[arrs[k].contiguous() for k in ARG_ORDER[scheme]])
    assert torch.equal(st, st2) and torch.equal(tally, tally2)
    # oracle on a random sample
    sel = torch.from_numpy(np.random.default_rng(5).choice(n, 2048, replace=False)).cuda()
    sample = {k: host(v[sel]) for k, v in arrs.items()}
    assert host(st[sel]).tolist() == oracle_verify(scheme, sample).tolist()
    # the blocking host-buffer call pipelines the batch in chunks of 2^18 items: ragged multi-chunk batch, same results
    nh = (1 << 19) + 777
    st_h, tally_h = eng.verify(scheme, *[host(arrs[k][:nh]) for k in ARG_ORDER[scheme]])
    assert (st_h == host(expect[:nh])).all()
    assert tally_h.tolist() == [int((expect[:nh] == k).sum()) for k in range(4)]


def test_public_key_derivation(eng, reference_kat):
    """PublicKey::from(&SecretKey): the reference's serde vectors give (sk, sk*G) and (sk*G, sk*G') for the same
    seeded sk (tests/serde.rs:34-81); random scalars against the oracle; a non-canonical scalar is flagged."""
    import torch
    from jubjub_schnorr_amd import serde
    v = reference_kat["serde_base58"]
    sk_ref = serde.b58decode(v["serde_secret_key"])
    rng = np.random.default_rng(8)
    sks = [int.from_bytes(sk_ref, "little"), 1, 2, o.R_ORDER - 1] + [int(x) for x in rng.integers(1, 1 << 62, 20)]
    sk = np.stack([fe_bytes(x) for x in sks] + [fe_bytes(o.R_ORDER)])            # last one: not canonical
    PK, PKp, bad = (host(t) for t in eng.public_keys(dev(sk), double=True))
    assert bad.tolist() == [0] * len(sks) + [1]
    for i, x in enumerate(sks):
        assert bytes(PK[i]) == pt_arr([o.mul(o.G, x)])[0].tobytes() and bytes(PKp[i]) == pt_arr([o.mul(o.G_NUMS, x)])[0].tobytes()
    comp = host(eng.compress(dev(np.concatenate([PK[:1], PKp[:1]]))))
    assert bytes(comp[0]) == serde.b58decode(v["serde_public_key"])
    assert bytes(comp[0]) + bytes(comp[1]) == serde.b58decode(v["serde_public_key_double"])
    PK1, bad1 = eng.public_keys(dev(sk))
    assert torch.equal(PK1.cpu(), torch.from_numpy(PK)) and host(bad1).tolist() == bad.tolist()


# ---- extended coordinates: what `PublicKey::verify` receives (reference src/keys/public.rs:114-118) ----------
@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
@pytest.mark.parametrize("n", [1, 65, 1000])
def test_verify_ext_matches_affine(eng, scheme, n):
    """Every point handed over as (U, V, Z) with a random Z: same statuses and tally as the oracle gives for the
    affine points; device-pointer and host-buffer entry points."""
    b = make_batch(scheme, n, seed=500 + n, n_keys=16)
    want = oracle_verify(scheme, b)
    ext = batch_to_extended(scheme, b, seed=n)
    st, tally = eng.verify_ext(scheme, *[dev(a) for a in ext])
    assert host(st).tolist() == want.tolist()
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]
    st_h, tally_h = eng.verify_ext(scheme, *ext)
    assert st_h.tolist() == want.tolist() and tally_h.tolist() == host(tally).tolist()


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_verify_ext_edge_cases(eng, scheme):
    """The adversarial items of the affine tests (small-order, off-curve, non-canonical ...) in extended form, plus
    what only this format can express: Z = 0 (InvalidPoint) and coordinates >= q (Malformed)."""
    b = edge_cases(scheme)
    want = oracle_verify(scheme, b).copy()
    ext = batch_to_extended(scheme, b, seed=9)
    names = ARG_ORDER[scheme]
    pts = [i for i, k in enumerate(names) if b[k].shape[1] == 64]
    # non-canonical affine coordinates cannot be rescaled: keep them verbatim with Z = 1 (still Malformed)
    for i in pts:
        raw = b[names[i]]
        big = np.array([to_int(r[:32]) >= o.Q or to_int(r[32:]) >= o.Q for r in raw])
        ext[i][big, :64] = raw[big]; ext[i][big, 64:] = fe_bytes(1)
    # valid item 0 repeated with Z = 0 on one point at a time, then with Z = q, then with U = q
    extra = []
    for i in pts:
        for kind in ("z0", "zq", "uq"):
            row = [a[0:1].copy() for a in ext]
            if kind == "z0":
                row[i][0, 64:] = 0
            elif kind == "zq":
                row[i][0, 64:] = fe_bytes(o.Q)
            else:
                row[i][0, :32] = fe_bytes(o.Q)
            extra.append((row, 1 if kind == "z0" else 3))
    assert want[0] == 0
    arrays = [np.concatenate([ext[j]] + [row[j] for row, _ in extra]) for j in range(len(ext))]
    want = np.concatenate([want, np.array([w for _, w in extra], np.uint8)])
    st, tally = eng.verify_ext(scheme, *[dev(a) for a in arrays])
    assert host(st).tolist() == want.tolist()
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]


def test_verify_ext_golden_vectors(eng):
    """The reference-held vectors (multisig KAT signature, serde signatures, legacy-double attack fixture) with
    their points rescaled by random Z."""
    from helpers import to_extended
    vec = json.load(open(os.path.join(GOLDEN, "verify_vectors.json")))
    H = lambda x: np.frombuffer(bytes.fromhex(x), np.uint8)  # noqa: E731
    rng = np.random.default_rng(2321)
    for scheme, items in vec.items():
        arrays = [np.stack([H(v[k]) for v in items]) for k in ARG_ORDER[scheme]]
        canonical = np.array([v["status"] != 3 for v in items])
        ext = [to_extended(a[canonical], rng, z_one_every=0) if a.shape[1] == 64 else a[canonical] for a in arrays]
        st, _ = eng.verify_ext(scheme, *ext)
        assert st.tolist() == [v["status"] for v in items if v["status"] != 3], scheme


@pytest.mark.parametrize("scheme", ["single", "double"])
def test_verify_ext_full_size(eng, scheme):
    """2^20 items in extended form built on the device (Z = a per-item value, U = u*Z, V = v*Z through the field
    multiplier): statuses known by construction, several items per lane share one inversion."""
    import torch
    import bench
    n = 1 << 20
    arrays, expect = bench.make_inputs(eng, scheme, n, 0)
    gen = torch.Generator(device="cpu").manual_seed(77)
    def to_ext(pts):
        z = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=gen); z[:, 31] &= 0x3F; z[:, 0] |= 1
        z = z.cuda()
        U = eng.debug_fq_mul(pts[:, :32].contiguous(), z); V = eng.debug_fq_mul(pts[:, 32:].contiguous(), z)
        return torch.cat([U, V, z], 1).contiguous()
    call = [to_ext(arrays[k]) if arrays[k].shape[1] == 64 else arrays[k] for k in ARG_ORDER[scheme]]
    st, tally = eng.verify_ext(scheme, *call)
    assert torch.equal(st, expect)
    assert host(tally).tolist() == [int((expect == k).sum()) for k in range(4)]


# ---- wire formats ---------------------------------------------------------------------
def test_decompress_and_compress(eng):
    from test_hostbuild import wire_point_cases
    enc = wire_point_cases(np.random.default_rng(41), n_random=256)
    arr = np.frombuffer(b"".join(enc), np.uint8).reshape(-1, 32)
    out, ok = eng.decompress(dev(arr))
    out, ok = host(out), host(ok)
    good = []
    for i, e in enumerate(enc):
        want = o.decompress(e)
        assert bool(ok[i]) == (want is not None), i
        assert out[i].tobytes() == ((o.le32(want[0]) + o.le32(want[1])) if want else (o.le32(0) + o.le32(1))), i
        if want:
            good.append(i)
    back = host(eng.compress(dev(out[good])))
    assert [r.tobytes() for r in back] == [enc[i] for i in good]


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
@pytest.mark.parametrize("n", [1, 65, 600])
def test_verify_wire(eng, scheme, n):
    """Wire entry points: same statuses as the affine path on decodable items; undecodable -> 3."""
    b = make_batch(scheme, n, seed=300 + n, n_keys=8)
    want = oracle_verify(scheme, b).copy()
    sig, pk, m = to_wire(scheme, b)
    # the mixed-order / small-order points of make_batch are on the curve, so they survive compression
    rng = np.random.default_rng(n)
    bad_rows = rng.choice(n, size=max(1, n // 10), replace=False)
    nonres = next(o.le32(v) for v in range(2, 100) if o.decompress(o.le32(v)) is None)
    for j, i in enumerate(bad_rows):
        kind = j % 3
        if kind == 0:
            sig[i, 32:64] = np.frombuffer(nonres, np.uint8)                 # R: no square root
        elif kind == 1:
            pk[i, :32] = np.frombuffer(o.le32(o.Q), np.uint8)                # PK: v = q
        else:
            z = bytearray(o.compress(o.IDENTITY)); z[31] |= 0x80             # PK: u = 0 with sign bit
            pk[i, :32] = np.frombuffer(bytes(z), np.uint8)
        want[i] = 3
    st, tally = eng.verify_wire(scheme, dev(sig), dev(pk), dev(m))
    assert host(st).tolist() == want.tolist()
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]
    st_h, tally_h = eng.verify_wire(scheme, sig, pk, m)       # blocking host-buffer entry point
    assert st_h.tolist() == want.tolist() and tally_h.tolist() == host(tally).tolist()


def test_verify_wire_golden_serde_bytes(eng, reference_kat):
    """The reference's own serialised signature / key bytes (tests/serde.rs, seed 2321) straight into the
    wire entry points."""
    v = reference_kat["serde_base58"]
    rng = o.StdRng(v["seed"]); rng.random_fr(); m = o.le32(rng.random_fq())
    A = lambda x, w: dev(np.frombuffer(x, np.uint8).reshape(1, w))  # noqa: E731
    st, _ = eng.verify_wire("single", A(o.b58decode(v["serde_signature"]), 64), A(o.b58decode(v["serde_public_key"]), 32), A(m, 32))
    assert host(st).tolist() == [0]
    st, _ = eng.verify_wire("double", A(o.b58decode(v["serde_signature_double"]), 96),
                            A(o.b58decode(v["serde_public_key_double"]), 64), A(m, 32))
    assert host(st).tolist() == [0]
    rng = o.StdRng(v["seed"]); rng.random_fr(); rng.random_fr(); m = o.le32(rng.random_fq())
    st, _ = eng.verify_wire("vargen", A(o.b58decode(v["serde_signature_var_gen"]), 64),
                            A(o.b58decode(v["serde_public_key_var_gen"]), 64), A(m, 32))
    assert host(st).tolist() == [0]
    k = reference_kat["multisig_kat"]
    st, _ = eng.verify_wire("single", A(bytes.fromhex(k["signature"]), 64), A(bytes.fromhex(k["aggregate_public_key"]), 32),
                            A(o.le32(k["message"]), 32))
    assert host(st).tolist() == [0]


# ---- multisig batch -------------------------------------------------------------------
def test_multisig_batch(eng, reference_kat):
    from test_hostbuild import check_multisig

    def run(z, PK, R, S, m, offs):
        out = eng.multisig_combine(dev(z), dev(PK), dev(R), dev(S), dev(m), offs)
        return tuple(host(t) for t in out)
    check_multisig(run, reference_kat)


def test_multisig_long_and_empty_transcripts(eng):
    """Transcripts of 257 and 1 000 participants (beyond the generated tag table: the tags are computed at call time,
    reference src/multisig.rs:326-338 takes any non-empty transcript), an empty transcript (status 5, alone) and ordinary
    ones in one call, everything `combine` returns against the oracle."""
    from test_hostbuild import check_long_multisig

    def run(z, PK, R, S, m, offs):
        out = eng.multisig_combine(dev(z), dev(PK), dev(R), dev(S), dev(m), offs)
        return tuple(host(t) for t in out)
    check_long_multisig(run, sizes=(257, 1000))


def test_multisig_many_transcripts(eng):
    """4096 copies of ragged oracle transcripts (different corruption pattern per copy is not needed:
    the point is many transcripts in flight and the per-transcript outputs staying separate)."""
    from helpers import make_multisig_batch
    z, PK, R, S, m, offs, want, info = make_multisig_batch(8, seed=33, max_n=6)
    reps = 512
    n = len(z)
    Z, P_, R_, S_, M = (np.tile(a, (reps, 1)) for a in (z, PK, R, S, m))
    offs_all = np.concatenate([[0]] + [offs[1:].astype(np.int64) + r * n for r in range(reps)]).astype(np.uint32)
    st, agg, su, sr, ts = (host(t) for t in eng.multisig_combine(dev(Z), dev(P_), dev(R_), dev(S_), dev(M), offs_all))
    assert (st.reshape(reps, n) == want[None, :]).all()
    assert (agg.reshape(reps, len(info), 64) == agg[:len(info)][None]).all()
    assert (sr.reshape(reps, len(info), 64) == sr[:len(info)][None]).all() and (su.reshape(reps, len(info), 32) == su[:len(info)][None]).all()
    for t, (a_pk, u, rsa) in enumerate(info):
        bad = want[offs[t]:offs[t + 1]].any()
        assert ts[t] == (4 if bad else 0) and su[t].tobytes() == (bytes(32) if bad else o.le32(u))


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_differential_32k_against_c_oracle(eng, scheme):
    """Every status byte and every challenge of a 2^15-item mixed batch (oracle-signed, oracle-corrupted:
    wrong keys, tampered messages, identity / order-2 / mixed-order points) against the C oracle."""
    n = 1 << 15
    b = make_batch(scheme, n, seed=4242, n_keys=512)
    # extra adversarial rows: R or PK moved into every torsion coset, off-curve points, non-canonical scalars
    t8 = torsion_generator()
    rng = np.random.default_rng(6)
    from helpers import pt_bytes, to_pt
    for k in range(1, 8):
        for name in ("R", "PK"):
            i = int(rng.integers(0, n))
            b[name][i] = pt_bytes(o.add(to_pt(b[name][i]), o.mul(t8, k)))
    for i in rng.integers(0, n, 8):
        b["R"][i, 32] ^= 1
    for i in rng.integers(0, n, 4):
        b["u"][i] = 0xFF
    want, want_c = oracle_verify(scheme, b, want_c=True)
    args = [dev(b[k]) for k in ARG_ORDER[scheme]]
    st, tally = eng.verify(scheme, *args)
    c = eng.challenge(scheme, *args[1:])
    got = host(st)
    assert (got == want).all(), np.where(got != want)[0][:10]
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]
    canonical = want != 3          # the oracle exports a challenge only for canonical inputs
    assert (host(c)[canonical] == want_c[canonical]).all()
    assert set(want.tolist()) == {0, 1, 2, 3}


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_key_table_path_against_oracle(eng, scheme):
    """2^17 items under 512 keys: the engine deduplicates the keys, builds per-key tables and verifies with additions
    only (csrc/key_tables.h).  Every status against the C oracle, with invalid keys shared by many items (identity,
    order 2, mixed order), wrong keys, tampered messages, small-order components on R and non-canonical scalars."""
    n = 1 << 17
    b = make_batch(scheme, n, seed=9090, n_keys=512)
    t8 = torsion_generator()
    rng = np.random.default_rng(16)
    from helpers import pt_bytes, to_pt
    for k in range(1, 8):
        i = int(rng.integers(0, n))
        b["R"][i] = pt_bytes(o.add(to_pt(b["R"][i]), o.mul(t8, k)))
    for i in rng.integers(0, n, 8):
        b["R"][i, 32] ^= 1                       # R off the curve
    for i in rng.integers(0, n, 4):
        b["u"][i] = 0xFF
    b["PK"][5, :32] = fe_bytes(o.Q)              # a non-canonical key
    b["R"][9] = pt_bytes(o.IDENTITY)
    want = oracle_verify(scheme, b)
    st, tally = eng.verify(scheme, *[dev(b[k]) for k in ARG_ORDER[scheme]])
    got = host(st)
    assert (got == want).all(), np.where(got != want)[0][:10]
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]
    assert set(want.tolist()) == {0, 1, 2, 3}
    # the blocking host-buffer entry points feed the same engine piece by piece (all columns of the first 2^16 items, then
    # the key columns of all the others, then the rest: csrc/host_calls.h run_host_block): the batch three times over
    # (3 x 2^17 items, so that every kind of piece occurs), affine and extended coordinates, same statuses
    tiled = {k: np.concatenate([v, v, v]) for k, v in b.items()}
    want3 = np.concatenate([want, want, want])
    before = eng.path_stats()
    st_h, tally_h = eng.verify(scheme, *[tiled[k] for k in ARG_ORDER[scheme]])
    assert (st_h == want3).all(), np.where(st_h != want3)[0][:10]
    assert tally_h.tolist() == [int((want3 == k).sum()) for k in range(4)]
    st_h, tally_h = eng.verify(scheme, *[b[k] for k in ARG_ORDER[scheme]])          # two pieces, every column in both
    assert (st_h == want).all() and tally_h.tolist() == host(tally).tolist()
    ext = [ext_on_device(eng, tiled[k]) if tiled[k].shape[1] == 64 else tiled[k] for k in ARG_ORDER[scheme]]
    st_e, tally_e = eng.verify_ext(scheme, *ext)
    assert (st_e == want3).all(), np.where(st_e != want3)[0][:10]
    assert tally_e.tolist() == [int((want3 == k).sum()) for k in range(4)]
    st_e, _ = eng.verify_ext(scheme, *[dev(a) for a in ext])
    assert (host(st_e) == want3).all()
    after = eng.path_stats()                              # all of these took the key tables (256+ signatures per key)
    assert after["key_tables_wide"] - before["key_tables_wide"] == 4, (before, after)
    assert after["throughput"] == before["throughput"] and after["keys_do_not_repeat"] == before["keys_do_not_repeat"]
    # the same items with every key made distinct in its bytes is impossible without re-signing; instead check the
    # fall-back decision: 2^16 items under 2^16 / 8 keys (8 per key: below the threshold) still verify the same
    m = 1 << 16
    sel = np.concatenate([np.arange(i, n, 512)[:8] for i in range(512)] * 16)[:m]
    sub = {k: v[sel] for k, v in b.items()}
    st2, _ = eng.verify(scheme, *[dev(sub[k]) for k in ARG_ORDER[scheme]])
    assert (host(st2) == want[sel]).all()


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_key_table_path_wire_against_oracle(eng, scheme):
    """The wire entry points on the key-table path: the 32-byte key encodings are deduplicated and every distinct key
    is decompressed once (csrc/key_tables.h kt_decode_key / kt_unpack_item).  2^17 items under 512 keys, every status
    against the C oracle: a key whose encoding is no point shared by all its items, a key replaced by its negative
    (sign bit) on all its items, single items with undecodable or non-canonical key bytes, undecodable R."""
    n = 1 << 17
    b = make_batch(scheme, n, seed=9191, n_keys=512)
    t8 = torsion_generator()
    rng = np.random.default_rng(17)
    from helpers import pt_bytes, to_pt
    for k in range(1, 8):
        i = int(rng.integers(0, n))
        b["R"][i] = pt_bytes(o.add(to_pt(b["R"][i]), o.mul(t8, k)))
    for i in rng.integers(0, n, 4):
        b["u"][i] = 0xFF
    key_col = "PKp" if scheme == "double" else ("Gen" if scheme == "vargen" else "PK")
    _, inverse = np.unique(b[key_col], axis=0, return_inverse=True)
    inverse = inverse.reshape(-1)
    def regular(i):        # an item whose key bytes are those of its key (item i uses key i mod 512), not a corrupted row
        while not ((b[key_col][i] == b[key_col][i + 512]).all() and (b[key_col][i] == b[key_col][i + 1024]).all()):
            i += 1
        return i
    i_neg, i_broken = regular(100), regular(7)
    negated = np.where(inverse == inverse[i_neg])[0]          # every item of one key: the key replaced by its negative
    pu, pv = to_pt(b[key_col][i_neg])
    b[key_col][negated] = pt_bytes(((o.Q - pu) % o.Q, pv))
    want = oracle_verify(scheme, b).copy()
    sig, pk, m = to_wire(scheme, b)
    nonres = np.frombuffer(next(o.le32(v) for v in range(2, 100) if o.decompress(o.le32(v)) is None), np.uint8)
    off = 0 if scheme == "single" else 32                     # the second key column of the two-column schemes
    broken = np.where(inverse == inverse[i_broken])[0]        # every item of another key: no square root
    assert len(broken) > 100 and len(negated) > 100
    pk[broken, off:off + 32] = nonres
    want[broken] = 3
    pk[11, :32] = np.frombuffer(o.le32(o.Q), np.uint8); want[11] = 3           # v = q on one item only
    z = bytearray(o.compress(o.IDENTITY)); z[31] |= 0x80
    pk[12, :32] = np.frombuffer(bytes(z), np.uint8); want[12] = 3              # u = 0 with the sign bit
    sig[13, 32:64] = nonres; want[13] = 3                                      # R undecodable
    st, tally = eng.verify_wire(scheme, dev(sig), dev(pk), dev(m))
    got = host(st)
    assert (got == want).all(), np.where(got != want)[0][:10]
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]
    assert set(want.tolist()) == {0, 1, 2, 3} and (want[negated] != 0).all()
    # host buffers, the batch three times over: the first ranges decode the keys of their own items, the others fetch their
    # key's point, decoded once for the whole call (double signatures: the key columns travel last and every range fetches)
    want3 = np.concatenate([want, want, want])
    st_h, tally_h = eng.verify_wire(scheme, *[np.concatenate([a, a, a]) for a in (sig, pk, m)])
    assert (st_h == want3).all(), np.where(st_h != want3)[0][:10]
    assert tally_h.tolist() == [int((want3 == k).sum()) for k in range(4)]
    st_h, _ = eng.verify_wire(scheme, sig, pk, m)
    assert (st_h == want).all()
    # below the key-table threshold (8 items per key) the keys are decoded item by item: same statuses
    sel = np.concatenate([np.arange(i, n, 512)[:8] for i in range(512)] * 16)[:1 << 16]
    st2, _ = eng.verify_wire(scheme, dev(sig[sel]), dev(pk[sel]), dev(m[sel]))
    assert (host(st2) == want[sel]).all()


@pytest.mark.parametrize("n", [(1 << 15) - 1, 1 << 15, 3 << 16, (3 << 16) + 1, 1 << 18, (1 << 18) + 1, (3 << 18) - 1, 3 << 18])
def test_key_kernel_scheduling_boundaries(eng, n):
    """Either side of the sizes at which a resident call orders its key kernels differently (csrc/verify_job.h: the key
    tables from 32 768 double signatures on, the keys counted ahead of the hashes above 3 * 2^16 and up to 2^18 items, the
    tables behind the hashes from 3 * 2^18): the bench mix under 4 096 keys, statuses known by construction."""
    import torch
    import bench
    for scheme in ("single", "double", "vargen"):
        arrays, expect = bench.make_inputs(eng, scheme, n, 0)
        for _ in range(2):                       # the second call finds the slot's memory of the first
            st, tally = eng.verify(scheme, *[arrays[k] for k in ARG_ORDER[scheme]])
            assert torch.equal(st, expect), (scheme, n)
            assert host(tally).tolist() == [int((expect == k).sum()) for k in range(4)]
        if n >= 1 << 18:
            # the same batches as extended points (normalised on the device ahead of the key kernels and the hashes)
            zgen = torch.Generator(device="cpu").manual_seed(n)

            def to_ext(pts):
                z = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=zgen)
                z[:, 31] &= 0x3F; z[:, 0] |= 1
                z = z.cuda()
                return torch.cat([eng.debug_fq_mul(pts[:, :32].contiguous(), z), eng.debug_fq_mul(pts[:, 32:].contiguous(), z), z], 1).contiguous()
            ext = [to_ext(arrays[k]) if arrays[k].shape[1] == 64 else arrays[k] for k in ARG_ORDER[scheme]]
            st, tally = eng.verify_ext(scheme, *ext)
            assert torch.equal(st, expect), (scheme, n, "ext")
            assert host(tally).tolist() == [int((expect == k).sum()) for k in range(4)]


@pytest.mark.parametrize("n_keys", [1, 2, 1024, 1025, 8191, 8192, 8193, 1 << 17])
def test_key_table_decision_boundary(eng, n_keys):
    """2^17 + 5 single signatures under 1 ... 2^17 keys: either side of the engine's on-device decisions (at most
    n / 128 distinct keys -> key tables with 6-bit windows, at most n / 16 -> 5-bit windows, else the throughput
    path), one key for every item (every lane on one hash slot), statuses by construction plus an oracle sample."""
    import torch
    n = (1 << 17) + 5
    gen = torch.Generator(device="cpu").manual_seed(4000 + n_keys)
    def scal(rows, top):
        t = torch.randint(0, 256, (rows, 32), dtype=torch.uint8, generator=gen)
        t[:, 31] &= top
        return t
    key_sk = scal(n_keys, 0x07); key_sk[:, 0] |= 1
    sk = key_sk[torch.arange(n) % n_keys].cuda()
    rnd, m = scal(n, 0x07).cuda(), scal(n, 0x3F).cuda()
    u, R, PK = eng.sign("single", sk, rnd, m)
    idx = torch.arange(n, device="cuda")
    expect = torch.zeros(n, dtype=torch.uint8, device="cuda")
    bad = (idx % 11) == 3
    m = m.clone(); m[bad, 0] ^= 1
    expect[bad] = 2
    st, tally = eng.verify("single", u, R, PK, m)
    assert torch.equal(st, expect)
    assert host(tally).tolist() == [int((expect == k).sum()) for k in range(4)]
    sel = torch.from_numpy(np.random.default_rng(n_keys).choice(n, 512, replace=False)).cuda()
    sample = {"u": host(u[sel]), "R": host(R[sel]), "PK": host(PK[sel]), "m": host(m[sel])}
    assert host(st[sel]).tolist() == oracle_verify("single", sample).tolist()


def test_key_table_pool_grows_with_the_keys_it_sees(eng):
    """2^20 single signatures under 32 768 keys (32 each: the key tables pay) need 4.1 GB of tables, more than the pool a
    call slot starts with: the first call takes the throughput path and says so in the path statistics, the next call
    finds the pool grown and takes the key tables; statuses identical (and right by construction) both times.  The
    SURVEY workload (4 096 keys) before it leaves the pool at its initial size: memory follows the keys, not the batch."""
    import torch
    import bench
    n = 1 << 20
    arrays, expect = bench.make_inputs(eng, "single", n, 0)
    call = [arrays[k] for k in ARG_ORDER["single"]]
    s = torch.cuda.Stream()                  # a stream of its own: the call takes a big slot no earlier test has used this way
    with torch.cuda.stream(s):
        st, _ = eng.verify("single", *call)
    s.synchronize()
    assert torch.equal(st, expect)
    small = eng.path_stats()
    arrays, expect = bench.make_inputs(eng, "single", n, 0, n_keys=32768)
    call = [arrays[k] for k in ARG_ORDER["single"]]
    with torch.cuda.stream(s):
        st1, t1 = eng.verify("single", *call)
    s.synchronize()
    mid = eng.path_stats()
    with torch.cuda.stream(s):
        st2, t2 = eng.verify("single", *call)
    s.synchronize()
    after = eng.path_stats()
    assert torch.equal(st1, expect) and torch.equal(st2, expect) and torch.equal(t1, t2)
    assert mid["keys_pool_too_small"] == small["keys_pool_too_small"] + 1, (small, mid)
    assert after["key_tables_narrow"] == mid["key_tables_narrow"] + 1, (mid, after)
    assert after["key_pool_bytes"] > mid["key_pool_bytes"] >= small["key_pool_bytes"] > 0
    assert after["key_pool_bytes"] - mid["key_pool_bytes"] < 6 << 30


def test_keys_crafted_to_collide_in_the_hash_table(eng):
    """2 000 distinct public-key byte strings built to land on ONE slot of the engine's key hash table as it would be
    with a known seed.  The product draws a fresh seed per call, so here they are ordinary keys: the call stays fast and
    every status is the oracle's.  tests/forcepath_child.py runs the same batch with the seed pinned (profiling build):
    the probe sequences are then cut at KT_MAX_PROBES and the batch takes the throughput path."""
    from helpers import crafted_collision_batch
    b = crafted_collision_batch(n_good=1 << 16)
    want = oracle_verify("single", b)
    import time
    import torch
    args = [dev(b[k]) for k in ARG_ORDER["single"]]
    eng.verify("single", *args)
    torch.cuda.synchronize()
    dt = 1.0
    for _ in range(3):            # the fastest of three: a pause of the test process itself is not the engine's
        t0 = time.perf_counter()
        st, tally = eng.verify("single", *args)
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    assert (host(st) == want).all()
    assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]
    assert dt < 0.05, dt          # ~5 ms on the throughput path; an unbounded chain of 2 000 keys would not matter yet,
    #                               a chain of 10^6 would: the bound is what keeps the worst case at this cost


def test_both_paths_at_every_size():
    """Throughput path and latency path forced in turn (profiling build, child process) on ragged sizes, edge
    cases and the torsion grid."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "forcepath_child.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "FORCEPATH OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("scheme,limit", [("single", 16384), ("double", 16384), ("vargen", 16384), ("single", 4096), ("double", 4096),
                                          ("vargen", 4096), ("single", 512), ("double", 1024), ("vargen", 256), ("single", 1024),
                                          ("vargen", 2048), ("single", 6144), ("double", 6144)])
def test_path_boundary(eng, scheme, limit):
    """Either side of the sizes at which the product changes method (csrc/engine_state.h SMALL_PATH_FINEST_ITEMS: 16 -> 8
    pieces on the latency path; SMALL_QUAD_CHAIN_MAX_ITEMS*: chains on quads -> on single lanes; SMALL_PATH_FINE_ITEMS: 8 -> 4
    pieces; SMALL_PATH_COOP_ITEMS: eight hash lanes -> one; SMALL_PATH_MAX_ITEMS: latency -> throughput path), against the oracle."""
    b = make_batch(scheme, limit + 1, seed=4711, n_keys=64)
    want = oracle_verify(scheme, b)
    for n in (limit, limit + 1):
        st, tally = eng.verify(scheme, *[dev(b[k][:n]) for k in ARG_ORDER[scheme]])
        assert (host(st) == want[:n]).all()
        assert host(tally).tolist() == [int((want[:n] == k).sum()) for k in range(4)]


def test_small_calls_on_different_streams_overlap_safely(eng):
    """Small calls take the engine's small slots in turn (disjoint buffers, no ordering between slots); with more
    streams than slots two calls share one and are ordered by its event.  Results must not depend on any of it."""
    import torch
    specs = [("single", 3000), ("double", 2000), ("vargen", 1500), ("single", 4096), ("double", 700), ("single", 1)]
    batches = [make_batch(s, n, seed=1200 + i, n_keys=16) for i, (s, n) in enumerate(specs)]
    args = [[dev(b[k]) for k in ARG_ORDER[s]] for b, (s, _) in zip(batches, specs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in specs]
    outs = []
    for _ in range(4):
        outs = []
        for a, (s, _), stream in zip(args, specs, streams):
            with torch.cuda.stream(stream):
                outs.append(eng.verify(s, *a))
    torch.cuda.synchronize()
    for (st, tally), b, (s, _) in zip(outs, batches, specs):
        want = oracle_verify(s, b)
        assert host(st).tolist() == want.tolist()
        assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]


def test_medium_calls_with_key_tables_on_different_streams(eng):
    """Calls of 65 536 ... 131 072 items take the medium slots in turn, each with its own workspace and key arena, and
    share the device's key stream: five key-table calls (one batch per scheme; affine and wire entry points) on five
    streams, more streams than slots, two rounds; every status against the oracle."""
    import torch
    batches = {"single": make_batch("single", 70000, seed=1500, n_keys=200), "double": make_batch("double", 66000, seed=1501, n_keys=200),
               "vargen": make_batch("vargen", 68000, seed=1502, n_keys=200)}
    want = {s: oracle_verify(s, b) for s, b in batches.items()}
    specs = [("single", False), ("double", False), ("vargen", True), ("single", True), ("double", True)]
    args = [[dev(a) for a in to_wire(s, batches[s])] if wire else [dev(batches[s][k]) for k in ARG_ORDER[s]] for s, wire in specs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in specs]
    outs = []
    for _ in range(2):
        outs = []
        for a, (s, wire), stream in zip(args, specs, streams):
            with torch.cuda.stream(stream):
                outs.append(eng.verify_wire(s, *a) if wire else eng.verify(s, *a))
    torch.cuda.synchronize()
    for (st, tally), (s, _) in zip(outs, specs):
        assert (host(st) == want[s]).all(), s
        assert host(tally).tolist() == [int((want[s] == k).sum()) for k in range(4)]


def test_calls_on_different_streams_do_not_interfere(eng):
    """The engine's workspaces are shared; launches from different streams must be ordered by the library."""
    import torch
    batches = [make_batch("single", 20000, seed=900 + i, n_keys=32) for i in range(2)] + [make_batch("vargen", 20000, seed=950, n_keys=32)]
    schemes = ["single", "single", "vargen"]
    args = [[dev(b[k]) for k in ARG_ORDER[s]] for b, s in zip(batches, schemes)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in batches]
    outs = []
    for _ in range(3):                       # interleave launches on three streams, several rounds
        outs = []
        for a, s, st in zip(args, schemes, streams):
            with torch.cuda.stream(st):
                outs.append(eng.verify(s, *a))
    torch.cuda.synchronize()
    for (st, tally), b, s in zip(outs, batches, schemes):
        want = oracle_verify(s, b)
        assert host(st).tolist() == want.tolist()
        assert host(tally).tolist() == [int((want == k).sum()) for k in range(4)]


def test_devcheck_device_against_host_stage_by_stage():
    """tools/devcheck: the same csrc/*.h functions on the device and on the host (field products, Hades rounds, point
    formulas, the pairing test, the Lehmer Euclid, the inversion by division steps against the power), raw limbs compared
    stage by stage -- what localises a miscompile or a device-only arithmetic difference."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jubjub_schnorr_amd", "tools", "devcheck")
    if not os.path.exists(exe):
        pytest.skip("tools/devcheck not built (python -c 'import __graft_entry__ as g; g.build()')")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "DEVCHECK OK" in p.stdout, p.stdout[-3000:] + p.stderr[-1000:]
    assert "inverse" in p.stdout
