#!/usr/bin/env python3
"""Build tests/golden/verify_vectors.json: boundary-level (affine, canonical bytes) verify vectors.

Inputs come from the reference's own tests (tests/golden/reference_kat.json: the multisig KAT
signature, the seed-2321 serde signatures of all three schemes, the legacy-double attack recipe,
and the sk = 0 behavioural cases of reference tests/schnorr*.rs); the expected status and challenge
are produced by oracle/jjs_oracle.py, which is itself pinned to those vectors
(tests/test_oracle_kat.py).  Runs anywhere (does not read /root/reference).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import jjs_oracle as o  # noqa: E402


def fe(x):
    return o.le32(x).hex()


def pt(p):
    return (o.le32(p[0]) + o.le32(p[1])).hex()


def main():
    kat = json.load(open(os.path.join(HERE, "reference_kat.json")))
    out = {"single": [], "double": [], "vargen": []}

    def add_single(name, u, R, PK, m):
        out["single"].append({"name": name, "u": fe(u) if u < 1 << 256 else None, "R": pt(R), "PK": pt(PK), "m": fe(m),
                              "status": o.verify_single(u, R, PK, m), "c": fe(o.challenge_single(R, PK, m))})

    def add_double(name, u, R, Rp, PK, PKp, m):
        out["double"].append({"name": name, "u": fe(u), "R": pt(R), "Rp": pt(Rp), "PK": pt(PK), "PKp": pt(PKp), "m": fe(m),
                              "status": o.verify_double(u, R, Rp, PK, PKp, m),
                              "c": fe(o.challenge_double(R, Rp, PK, PKp, m))})

    def add_vargen(name, u, R, PK, Gen, m):
        out["vargen"].append({"name": name, "u": fe(u), "R": pt(R), "PK": pt(PK), "Gen": pt(Gen), "m": fe(m),
                              "status": o.verify_vargen(u, R, PK, Gen, m), "c": fe(o.challenge_vargen(R, PK, Gen, m))})

    # multisig KAT aggregate signature (reference src/multisig.rs:625-672), verifies via PublicKey::verify
    k = kat["multisig_kat"]
    sig = bytes.fromhex(k["signature"])
    R, PK = o.decompress(sig[32:]), o.decompress(bytes.fromhex(k["aggregate_public_key"]))
    add_single("multisig_kat_aggregate", o.from_le(sig[:32]), R, PK, k["message"])
    assert out["single"][-1]["c"] == k["challenge"] and out["single"][-1]["status"] == 0

    # serde vectors, seed 2321 (reference tests/serde.rs)
    seed = kat["serde_base58"]["seed"]
    rng = o.StdRng(seed); sk = rng.random_fr(); m = rng.random_fq(); u, R = o.sign_single(rng, sk, m)
    pk, pkp = o.mul(o.G, sk), o.mul(o.G_NUMS, sk)
    add_single("serde_signature", u, R, pk, m)
    add_single("serde_signature_wrong_key", u, R, o.mul(o.G, sk + 1), m)          # tests/schnorr.rs:29-44
    add_single("serde_signature_identity_pk", u, R, o.IDENTITY, m)                 # tests/schnorr.rs:58-66
    add_single("serde_signature_order2_pk", u, R, o.ORDER2, m)
    add_single("serde_signature_mixed_order_R", u, o.add(R, o.ORDER2), pk, m)
    add_single("serde_signature_tampered_m", u, R, pk, (m + 1) % o.Q)
    rng = o.StdRng(seed); sk = rng.random_fr(); m = rng.random_fq(); u, R, Rp = o.sign_double(rng, sk, m)
    add_double("serde_signature_double", u, R, Rp, pk, pkp, m)
    add_double("serde_signature_double_identity", u, R, Rp, o.IDENTITY, o.IDENTITY, m)   # tests/schnorr_double.rs:61-69
    add_double("serde_signature_double_wrong_pk_prime", u, R, Rp, pk, o.mul(o.G_NUMS, sk + 1), m)
    rng = o.StdRng(seed); sk = rng.random_fr(); g = rng.random_fr(); m = rng.random_fq()
    gen = o.mul(o.G, g); pkv = o.mul(gen, sk); u, R = o.sign_vargen(rng, sk, gen, m)
    add_vargen("serde_signature_var_gen", u, R, pkv, gen, m)
    add_vargen("serde_signature_var_gen_identity", u, R, o.IDENTITY, gen, m)             # tests/schnorr_var_generator.rs:116-124
    add_vargen("serde_signature_var_gen_other_generator", u, R, pkv, o.mul(o.G, g + 1), m)  # :61-113 shape

    # legacy double attack (reference tests/common/mod.rs:23-66) -> InvalidSignature
    a = kat["legacy_double_attack"]
    sk, m, nonce = a["sk"], a["message"], a["nonce"]
    pk = o.mul(o.G, sk); r = o.mul(o.G, nonce); rp = o.mul(o.G_NUMS, a["r_prime_scalar"])
    lc = o.digest_truncated([r[0], r[1], rp[0], rp[1], pk[0], pk[1], m])
    u = (nonce - lc * sk) % o.R_ORDER
    pkp = o.mul(o.add(rp, o.neg(o.mul(o.G_NUMS, u))), pow(lc, -1, o.R_ORDER))
    add_double("legacy_double_attack", u, r, rp, pk, pkp, m)
    assert out["double"][-1]["status"] == a["expected_status"]

    # cross-generator forgery (reference tests/schnorr_var_generator.rs:61-113, seed 0xdead): expected
    # InvalidSignature; under the generator-free legacy challenge the forged pair would verify
    rng = o.StdRng(0xDEAD)
    sk = rng.random_fr(); g = rng.random_fr(); gen = o.mul(o.G, g); pkv = o.mul(gen, sk)
    m = rng.random_fq()
    u, R = o.sign_vargen(rng, sk, gen, m)
    add_vargen("cross_generator_original", u, R, pkv, gen, m)
    assert out["vargen"][-1]["status"] == 0
    p_pt = o.mul(o.G, rng.random_fr())
    c_prime = o.digest_truncated([R[0], R[1], p_pt[0], p_pt[1], m])
    g2 = o.mul(o.add(R, o.neg(o.mul(p_pt, c_prime))), pow(u, -1, o.R_ORDER))
    assert o._equation(g2, u, p_pt, c_prime, R)          # the attack works against the legacy transcript
    add_vargen("cross_generator_forgery", u, R, p_pt, g2, m)
    assert out["vargen"][-1]["status"] == 2
    # sk = 0 with seeded messages (tests/schnorr.rs:58-66 seed 0xbeef-style cases): identity keys
    for scheme, seed in (("single", 0xBEEF), ("double", 0xBEEF), ("vargen", 0xBEEF)):
        rng = o.StdRng(seed)
        m = rng.random_fq()
        if scheme == "single":
            u, R = o.sign_single(rng, 0, m)
            add_single("sk_zero_identity_pk", u, R, o.IDENTITY, m)
            assert out["single"][-1]["status"] == 1
        elif scheme == "double":
            u, R, Rp = o.sign_double(rng, 0, m)
            add_double("sk_zero_identity_pk", u, R, Rp, o.IDENTITY, o.IDENTITY, m)
            assert out["double"][-1]["status"] == 1
        else:
            u, R = o.sign_vargen(rng, 0, o.G, m)
            add_vargen("sk_zero_identity_pk", u, R, o.IDENTITY, o.G, m)
            assert out["vargen"][-1]["status"] == 1
    # challenge binds both public keys (reference src/signatures/double.rs:190-217)
    r, rp, m13 = o.mul(o.G, 11), o.mul(o.G_NUMS, 11), 13
    pk17, pkp17 = o.mul(o.G, 17), o.mul(o.G_NUMS, 17)
    base_c = o.challenge_double(r, rp, pk17, pkp17, m13)
    assert base_c != o.challenge_double(r, rp, o.add(pk17, o.G), pkp17, m13)
    assert base_c != o.challenge_double(r, rp, pk17, o.add(pkp17, o.G_NUMS), m13)
    add_double("challenge_binds_pk_base", 1, r, rp, pk17, pkp17, m13)
    add_double("challenge_binds_pk_changed_pk", 1, r, rp, o.add(pk17, o.G), pkp17, m13)
    add_double("challenge_binds_pk_changed_pk_prime", 1, r, rp, pk17, o.add(pkp17, o.G_NUMS), m13)
    assert len({v["c"] for v in out["double"][-3:]}) == 3

    with open(os.path.join(HERE, "verify_vectors.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print({k: [(v["name"], v["status"]) for v in vs] for k, vs in out.items()})


if __name__ == "__main__":
    main()
