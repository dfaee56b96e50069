"""Pin the Python oracle against every byte-level vector the reference's own tests hold.

Vectors: tests/golden/reference_kat.json (extracted by tests/golden/make_reference_kat.py
from reference src/multisig.rs:544-672 and tests/serde.rs:34-142).  SURVEY.md Appendix A
check values for the Poseidon parameterisation are asserted too.
"""
import hashlib

import jjs_oracle as o


def H(x):
    return bytes.fromhex(x)


def test_poseidon_constants_check_values():
    # SURVEY.md A.3 item 5
    assert o.RC_PLAIN[0] == 0x4929E824CAE3E5B6915AF89C2B2EF56233518DA79404494933A12BB7322DD247
    assert o.RC[0] == 0x6D67DFB07C22C6FD0B22407B580659556E7C8F8B712CAB9E973D2BB834DE71C5
    assert o.RC[1] == 0x0B9CCE2D2D53F55BC6B43FF40F0D5027EDF6F79D698F81837CCD0775CDE2CB36
    assert o.RC[339] == 0x33FAD9B52943648E1098BBFFA93E758268B8EB565DA6C8390BF6E77A839652ED
    assert o.MDS[0][0] == 0x04D4237855C1011651E8DCC995BF433111B424CB999A419A0000000066666666
    assert o.MDS[0][1] == 0x514F37C663347F1811134802ADE09D571BE9E1565554C954FFFFFFFFAAAAAAAB
    assert o.MDS[4][4] == 0x521D817B8C8FE0FF7E0B745318E0FE2A40C8936413B058EBC4EC4EC462762763
    blob = b"".join(o.le32(c) for c in o.RC) + b"".join(o.le32(o.MDS[i][j]) for i in range(5) for j in range(5))
    assert hashlib.sha256(blob).hexdigest() == "f1caa4c7cdb9e9dfc95ef8d6f66288e629556393d6930da3584dc4d15b876adf"
    assert o.sponge_tag(5) == 0x4A160E2860BF61DBE4F2307D562BC8B987B234208740A3C8DA695AA49D726B0E
    assert o.sponge_tag(7) == 0x519C1A9313D7C1017DAF42055603D0A1B0A38BB4E1B353C1F4ABBBD515044145
    assert o.sponge_tag(10) == 0x26D1663A7872E55B70AB67864B6C2AF938EB73E997313BB626CBB554D950CFC7


def test_curve_constants():
    assert o.is_on_curve(o.G) and o.is_on_curve(o.G_NUMS)
    assert o.is_torsion_free(o.G) and o.is_torsion_free(o.G_NUMS)
    assert o.compress(o.G).hex() == "12" + "00" * 31
    assert o.compress(o.G_NUMS).hex() == "f83e2e1607b705677a50a5820fba4999fd343bebbe2d167b1bebf3b2b30ed8c3"
    assert o.compress(o.ORDER2).hex() == "00000000fffffffffe5bfeff02a4bd5305d8a10908d83933487d9d2953a7ed73"
    assert o.le32(o.R_ORDER).hex() == "b72cf7d65e0e97d08210c8cc932068a6003b3401013b6706a9af3365eab47d0e"
    assert o.decompress(o.compress(o.G_NUMS)) == o.G_NUMS


def test_multisig_known_answer(reference_kat):
    k = reference_kat["multisig_kat"]
    sks, rs, ss, m = k["secret_keys"], k["r_scalars"], k["s_scalars"], k["message"]
    pks = [o.mul(o.G, s) for s in sks]
    Rs = [o.mul(o.G, s) for s in rs]
    Ss = [o.mul(o.G, s) for s in ss]
    assert [o.compress(p).hex() for p in pks] == k["public_keys"]
    assert [o.compress(p).hex() for p in Rs] == k["r_points"]
    assert [o.compress(p).hex() for p in Ss] == k["s_points"]
    ds, agg, a, rsa, c = o.multisig_transcript(pks, Rs, Ss, m)
    assert [o.le32(d).hex() for d in ds] == k["delinearization"]          # three 8-input hashes
    assert o.compress(agg).hex() == k["aggregate_public_key"]
    assert o.le32(a).hex() == k["binding_coefficient"]                     # 15-input hash
    assert o.compress(rsa).hex() == k["aggregate_commitment"]
    assert o.le32(c).hex() == k["challenge"]                               # 5-input standard challenge
    shares = [(rs[i] + ss[i] * a - c * ds[i] * sks[i]) % o.R_ORDER for i in range(3)]
    assert [o.le32(z).hex() for z in shares] == k["individual_shares"]
    u = sum(shares) % o.R_ORDER
    sig = o.le32(u) + o.compress(rsa)
    assert sig.hex() == k["signature"]
    # the aggregate verifies through PublicKey::verify (src/multisig.rs:90-92)
    assert o.challenge_single(rsa, agg, m) == c
    assert o.verify_single(u, rsa, agg, m) == o.OK
    # the wire form decodes to the same points
    assert o.decompress(H(k["signature"])[32:]) == rsa
    assert o.decompress(H(k["aggregate_public_key"])) == agg


def test_serde_vectors_seed_2321(reference_kat):
    v = reference_kat["serde_base58"]
    seed = v["seed"]

    rng = o.StdRng(seed)
    sk = rng.random_fr()
    assert o.le32(sk) == o.b58decode(v["serde_secret_key"])
    pk = o.mul(o.G, sk)
    assert o.compress(pk) == o.b58decode(v["serde_public_key"])
    pkp = o.mul(o.G_NUMS, sk)
    assert o.compress(pk) + o.compress(pkp) == o.b58decode(v["serde_public_key_double"])

    # Signature: sk, msg, sign
    rng = o.StdRng(seed)
    sk = rng.random_fr()
    m = rng.random_fq()
    u, R = o.sign_single(rng, sk, m)
    assert o.le32(u) + o.compress(R) == o.b58decode(v["serde_signature"])
    assert o.verify_single(u, R, pk, m) == o.OK

    # SignatureDouble
    rng = o.StdRng(seed)
    sk = rng.random_fr()
    m = rng.random_fq()
    u, R, Rp = o.sign_double(rng, sk, m)
    assert o.le32(u) + o.compress(R) + o.compress(Rp) == o.b58decode(v["serde_signature_double"])
    assert o.verify_double(u, R, Rp, pk, pkp, m) == o.OK

    # var-gen key material: sk then generator scalar (src/keys/secret/var_gen.rs:166-168)
    rng = o.StdRng(seed)
    sk = rng.random_fr()
    g = rng.random_fr()
    gen = o.mul(o.G, g)
    assert o.le32(sk) + o.compress(gen) == o.b58decode(v["serde_secret_key_var_gen"])
    pkv = o.mul(gen, sk)
    assert o.compress(pkv) + o.compress(gen) == o.b58decode(v["serde_public_key_var_gen"])
    m = rng.random_fq()
    u, R = o.sign_vargen(rng, sk, gen, m)
    assert o.le32(u) + o.compress(R) == o.b58decode(v["serde_signature_var_gen"])
    assert o.verify_vargen(u, R, pkv, gen, m) == o.OK


def test_error_classes_match_reference_behaviour(reference_kat):
    rng = o.StdRng(2321)
    sk = rng.random_fr()
    m = rng.random_fq()
    u, R = o.sign_single(rng, sk, m)
    pk = o.mul(o.G, sk)
    wrong_pk = o.mul(o.G, rng.random_fr())
    # tests/schnorr.rs:29-44 wrong key -> InvalidSignature
    assert o.verify_single(u, R, wrong_pk, m) == o.INVALID_SIGNATURE
    # tests/schnorr.rs:58-66 sk = 0 -> identity PK -> InvalidPoint
    assert o.verify_single(u, R, o.IDENTITY, m) == o.INVALID_POINT
    # small order / mixed order / off curve -> InvalidPoint
    assert o.verify_single(u, R, o.ORDER2, m) == o.INVALID_POINT
    assert o.verify_single(u, R, o.add(pk, o.ORDER2), m) == o.INVALID_POINT
    assert o.verify_single(u, o.add(R, o.ORDER2), pk, m) == o.INVALID_POINT
    assert o.verify_single(u, R, (pk[0], (pk[1] + 1) % o.Q), m) == o.INVALID_POINT
    # non canonical
    assert o.verify_single(o.R_ORDER, R, pk, m) == o.MALFORMED
    assert o.verify_single(u, R, pk, o.Q) == o.MALFORMED
    # tampered message
    assert o.verify_single(u, R, pk, (m + 1) % o.Q) == o.INVALID_SIGNATURE

    # double, sk = 0 (tests/schnorr_double.rs:61-69)
    u, R, Rp = o.sign_double(rng, sk, m)
    assert o.verify_double(u, R, Rp, o.IDENTITY, o.IDENTITY, m) == o.INVALID_POINT
    assert o.verify_double(u, R, Rp, wrong_pk, o.mul(o.G_NUMS, sk), m) == o.INVALID_SIGNATURE

    # var-gen, cross-generator forgery shape (tests/schnorr_var_generator.rs:61-113)
    gen = o.mul(o.G, 5)
    u, R = o.sign_vargen(rng, sk, gen, m)
    assert o.verify_vargen(u, R, o.mul(gen, sk), gen, m) == o.OK
    gen2 = o.mul(o.G, 7)
    assert o.verify_vargen(u, R, o.mul(gen, sk), gen2, m) == o.INVALID_SIGNATURE
    assert o.verify_vargen(u, R, o.IDENTITY, gen, m) == o.INVALID_POINT


def legacy_double_fixture(reference_kat):
    """tests/common/mod.rs:23-66 recipe."""
    k = reference_kat["legacy_double_attack"]
    sk, m, nonce = k["sk"], k["message"], k["nonce"]
    pk = o.mul(o.G, sk)
    r = o.mul(o.G, nonce)
    rp = o.mul(o.G_NUMS, k["r_prime_scalar"])
    legacy_c = o.digest_truncated([r[0], r[1], rp[0], rp[1], pk[0], pk[1], m])
    inv = pow(legacy_c, -1, o.R_ORDER)
    u = (nonce - legacy_c * sk) % o.R_ORDER
    pkp = o.mul(o.add(rp, o.neg(o.mul(o.G_NUMS, u))), inv)
    return u, r, rp, pk, pkp, m, legacy_c


def test_legacy_double_attack_is_rejected(reference_kat):
    u, r, rp, pk, pkp, m, legacy_c = legacy_double_fixture(reference_kat)
    # SURVEY.md B.4 bytes (re-derived here)
    assert o.le32(u).hex() == "b9f3dc4321caaca2c32c6b33718bae703c3845a1889be18b9535f1ec12465f0a"
    assert o.compress(r).hex() == "80da3101692e3ed7208f88ddca700ab6acb35fa21e8e83e50d212bde3568af3f"
    assert o.compress(rp).hex() == "2859a98c0dbcd98bf578e1602627b0c107be24a05ac5474edea512b44f06d901"
    assert o.compress(pk).hex() == "030fa17156c36af83deba1a959c722cef3ae9c5c23a1ee36aa5f22175cd4b3b2"
    assert o.compress(pkp).hex() == "3e1dd46bae1f94b2ab35e3f89c48a2fe8f54f2d3497a10d25f77bf2b8555fb91"
    for p in (r, rp, pk, pkp):
        assert o.point_is_valid(p)
    # under the legacy 7-element transcript both equations hold ...
    assert o._equation(o.G, u, pk, legacy_c, r) and o._equation(o.G_NUMS, u, pkp, legacy_c, rp)
    # ... the real verifier rejects (tests/schnorr_double.rs:72-82)
    assert o.verify_double(u, r, rp, pk, pkp, m) == reference_kat["legacy_double_attack"]["expected_status"]


def test_multisig_shares_and_combine(reference_kat):
    """verify_share / combine / sign_round_2 restatements against the multisig KAT."""
    k = reference_kat["multisig_kat"]
    sks, rs, ss, m = k["secret_keys"], k["r_scalars"], k["s_scalars"], k["message"]
    pks = [o.mul(o.G, s) for s in sks]; Rs = [o.mul(o.G, s) for s in rs]; Ss = [o.mul(o.G, s) for s in ss]
    zs = [o.multisig_sign_share(sks[i], rs[i], ss[i], pks, Rs, Ss, m) for i in range(3)]
    assert [o.le32(z).hex() for z in zs] == k["individual_shares"]
    assert all(o.multisig_verify_share(zs[i], i, pks, Rs, Ss, m) for i in range(3))
    assert not o.multisig_verify_share(zs[0], 1, pks, Rs, Ss, m)
    assert not o.multisig_verify_share((zs[2] + 1) % o.R_ORDER, 2, pks, Rs, Ss, m)
    (u, rsa), bad = o.multisig_combine(zs, pks, Rs, Ss, m)
    assert bad is None and (o.le32(u) + o.compress(rsa)).hex() == k["signature"]
    assert o.multisig_combine([zs[0], zs[2], zs[1]], pks, Rs, Ss, m) == (None, 1)
