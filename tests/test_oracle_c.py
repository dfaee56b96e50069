"""Pin the C restatement: reference KATs first, then agreement with the Python oracle."""
import numpy as np
import pytest

import jjs_oracle as o
import jjs_oracle_c as oc
from helpers import (ARG_ORDER, edge_cases, fe_arr, fe_bytes, make_batch, oracle_verify, pt_arr, pt_bytes,
                     py_verify, rand_mod, to_int, to_pt, torsion_generator)


def H(x):
    return np.frombuffer(bytes.fromhex(x), np.uint8)


def test_kat_single_verify_and_challenge(reference_kat):
    """the multisig KAT's aggregate signature through PublicKey::verify."""
    k = reference_kat["multisig_kat"]
    sig = bytes.fromhex(k["signature"])
    R = o.decompress(sig[32:])
    PK = o.decompress(bytes.fromhex(k["aggregate_public_key"]))
    st, c = oc.verify_single(H(k["signature"])[None, :32], pt_arr([R]), pt_arr([PK]), fe_arr([k["message"]]),
                             want_c=True)
    assert st[0] == 0
    assert c[0].tobytes().hex() == k["challenge"]


def test_kat_fixed_base_points(reference_kat):
    k = reference_kat["multisig_kat"]
    scal = k["secret_keys"] + k["r_scalars"] + k["s_scalars"]
    want = k["public_keys"] + k["r_points"] + k["s_points"]
    out = oc.scalar_mul(np.tile(pt_bytes(o.G), (9, 1)), fe_arr(scal))
    assert [o.compress(to_pt(r)).hex() for r in out] == want


def test_kat_poseidon_8_and_15_inputs(reference_kat):
    k = reference_kat["multisig_kat"]
    pks = [o.decompress(bytes.fromhex(x)) for x in k["public_keys"]]
    pre = []
    for pk in pks:
        row = [pk[0], pk[1]]
        for p in pks:
            row += [p[0], p[1]]
        pre.append(fe_arr(row))
    out = oc.poseidon(np.stack(pre))
    got = [(to_int(r) & ((1 << 250) - 1)) for r in out]
    assert [o.le32(x).hex() for x in got] == k["delinearization"]
    agg = o.decompress(bytes.fromhex(k["aggregate_public_key"]))
    row = [agg[0], agg[1], k["message"]]
    for rp, sp in zip(k["r_points"], k["s_points"]):
        a, b = o.decompress(bytes.fromhex(rp)), o.decompress(bytes.fromhex(sp))
        row += [a[0], a[1], b[0], b[1]]
    out = oc.poseidon(fe_arr(row)[None])
    assert o.le32(to_int(out[0]) & ((1 << 250) - 1)).hex() == k["binding_coefficient"]


def test_serde_vectors_through_c(reference_kat):
    v = reference_kat["serde_base58"]
    rng = o.StdRng(v["seed"])
    sk = rng.random_fr(); m = rng.random_fq(); rnd = rng.random_fr()
    u, R, PK = oc.sign_single(fe_arr([sk]), fe_arr([rnd]), fe_arr([m]))
    assert u[0].tobytes() + o.compress(to_pt(R[0])) == o.b58decode(v["serde_signature"])
    assert o.compress(to_pt(PK[0])) == o.b58decode(v["serde_public_key"])
    assert oc.verify_single(u, R, PK, fe_arr([m]))[0] == 0
    u, R, Rp, PK, PKp = oc.sign_double(fe_arr([sk]), fe_arr([rnd]), fe_arr([m]))
    assert u[0].tobytes() + o.compress(to_pt(R[0])) + o.compress(to_pt(Rp[0])) == o.b58decode(v["serde_signature_double"])
    assert o.compress(to_pt(PK[0])) + o.compress(to_pt(PKp[0])) == o.b58decode(v["serde_public_key_double"])
    assert oc.verify_double(u, R, Rp, PK, PKp, fe_arr([m]))[0] == 0
    rng = o.StdRng(v["seed"])
    sk = rng.random_fr(); g = rng.random_fr(); m = rng.random_fq(); rnd = rng.random_fr()
    u, R, PK, Gen = oc.sign_vargen(fe_arr([sk]), fe_arr([g]), fe_arr([rnd]), fe_arr([m]))
    assert u[0].tobytes() + o.compress(to_pt(R[0])) == o.b58decode(v["serde_signature_var_gen"])
    assert o.compress(to_pt(PK[0])) + o.compress(to_pt(Gen[0])) == o.b58decode(v["serde_public_key_var_gen"])
    assert oc.verify_vargen(u, R, PK, Gen, fe_arr([m]))[0] == 0


def test_legacy_double_attack_through_c(reference_kat):
    from test_oracle_kat import legacy_double_fixture
    u, r, rp, pk, pkp, m, _ = legacy_double_fixture(reference_kat)
    st = oc.verify_double(fe_arr([u]), pt_arr([r]), pt_arr([rp]), pt_arr([pk]), pt_arr([pkp]), fe_arr([m]))
    assert st[0] == reference_kat["legacy_double_attack"]["expected_status"] == 2


def test_field_and_hash_primitives_match_python():
    rng = np.random.default_rng(5)
    a, b = rand_mod(rng, 64, o.Q), rand_mod(rng, 64, o.Q)
    a[0] = fe_bytes(o.Q - 1); b[0] = fe_bytes(o.Q - 1); a[1] = fe_bytes(0)
    out = oc.fq_mul(a, b)
    for i in range(64):
        assert to_int(out[i]) == to_int(a[i]) * to_int(b[i]) % o.Q
    for k in (1, 4, 5, 7, 8, 10, 15):
        x = rand_mod(rng, 3 * k, o.Q).reshape(3, k, 32)
        out = oc.poseidon(x)
        for i in range(3):
            assert to_int(out[i]) == o.poseidon_digest([to_int(r) for r in x[i]])


def test_point_flags_all_cosets():
    t8 = torsion_generator()
    s = o.mul(o.G, 123456789)
    pts = [o.add(s, o.mul(t8, k)) for k in range(8)] + [o.mul(t8, k) for k in range(8)]
    flags = oc.point_flags(pt_arr(pts))
    for i, p in enumerate(pts):
        want = int(o.is_on_curve(p)) | (int(o.is_torsion_free(p)) << 1) | (int(o.is_identity(p)) << 2)
        assert flags[i] == want
    assert list(flags[:8] >> 1 & 1) == [1, 0, 0, 0, 0, 0, 0, 0]


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_statuses_match_python_on_mixed_batch(scheme):
    b = make_batch(scheme, 96, seed=11, n_keys=8)
    st, c = oracle_verify(scheme, b, want_c=True)
    assert (st == 0).sum() > 48 and len(set(st.tolist())) >= 2
    for i in range(len(st)):
        assert st[i] == py_verify(scheme, b, i), i


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_statuses_match_python_on_edge_cases(scheme):
    b = edge_cases(scheme)
    st = oracle_verify(scheme, b)
    want = [py_verify(scheme, b, i) for i in range(len(st))]
    assert st.tolist() == want
    assert set(want) == {0, 1, 2, 3}


def test_empty_batch():
    z32, z64 = np.zeros((0, 32), np.uint8), np.zeros((0, 64), np.uint8)
    assert len(oc.verify_single(z32, z64, z64, z32)) == 0


def test_multisig_combine_port_against_the_reference_kat(reference_kat):
    """jjo_multisig_combine (the timed CPU baseline of the multisig batch: the reference's combine / verify_share algorithm,
    src/multisig.rs:326-387, 393-500) reproduces the reference's multisig KAT bytes and the Python oracle on ragged transcripts."""
    from test_hostbuild import check_multisig

    def run(z, PK, R, S, m, offs):
        share, tst, agg, su, sr = oc.multisig_combine(z, PK, R, S, m, offs)
        return share, agg, su, sr, tst
    check_multisig(run, reference_kat)
    # an empty transcript between two others: status 5 for itself only
    from helpers import make_multisig_batch
    z, PK, R, S, m, offs, want, info = make_multisig_batch(2, seed=3, corrupt=False)
    offs3 = np.array([0, offs[1], offs[1], offs[2]], np.uint32)
    m3 = np.stack([m[0], m[0], m[1]])
    share, tst, agg, su, sr = oc.multisig_combine(z, PK, R, S, m3, offs3)
    assert tst.tolist() == [0, 5, 0] and not agg[1].any() and share.tolist() == [0] * len(z)
