"""Host-side ingest of the reference's serde strings (base58 of to_bytes(); reference tests/serde.rs)."""
import json
import os

import numpy as np
import pytest

import jjs_oracle as o
from jubjub_schnorr_amd import serde


def test_base58_matches_the_reference_strings(reference_kat):
    v = reference_kat["serde_base58"]
    sizes = {"serde_public_key": 32, "serde_public_key_double": 64, "serde_public_key_var_gen": 64, "serde_secret_key": 32,
             "serde_secret_key_var_gen": 64, "serde_signature": 64, "serde_signature_double": 96, "serde_signature_var_gen": 64}
    for name, size in sizes.items():
        raw = serde.b58decode(v[name])
        assert len(raw) == size and raw == o.b58decode(v[name])
        assert serde.b58encode(raw) == v[name]
    # pk = sk * G, the relation the reference's vectors carry (tests/serde.rs:34-62)
    sk = int.from_bytes(serde.b58decode(v["serde_secret_key"]), "little")
    assert o.compress(o.mul(o.G, sk)) == serde.b58decode(v["serde_public_key"])


def test_base58_round_trip_and_leading_zeros():
    rng = np.random.default_rng(3)
    for n in (1, 31, 32, 64, 96):
        for _ in range(20):
            b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
            assert serde.b58decode(serde.b58encode(b)) == b
    assert serde.b58encode(b"\x00\x00\x01") == "112" and serde.b58decode("112") == b"\x00\x00\x01"
    assert serde.b58decode("") == b""


def test_bad_strings_fail_like_the_reference_deserialiser(reference_kat):
    v = reference_kat["serde_base58"]
    with pytest.raises(serde.SerdeError) as e:
        serde.decode_column([v["serde_public_key"], v["serde_public_key"][:-1] + "0"], "PublicKey")   # '0' is not base58
    assert e.value.index == 1
    with pytest.raises(serde.SerdeError) as e:
        serde.decode_column([v["serde_signature"]], "PublicKey")                                      # 64 bytes, not 32
    assert "invalid length 64" in e.value.reason
    assert serde.decode_column([], "Signature").shape == (0, 64)


@pytest.mark.gpu
def test_verify_from_serde_strings(reference_kat):
    import jubjub_schnorr_amd as jjs
    eng = jjs.engine()
    v = reference_kat["serde_base58"]
    rng = o.StdRng(v["seed"]); rng.random_fr(); m = o.le32(rng.random_fq())
    msg = np.frombuffer(m, np.uint8).reshape(1, 32)
    st, tally = serde.verify_strings(eng, "single", [v["serde_signature"]], [v["serde_public_key"]], msg)
    assert st.tolist() == [0] and tally.tolist() == [1, 0, 0, 0]
    st, _ = serde.verify_strings(eng, "double", [v["serde_signature_double"]], [v["serde_public_key_double"]], msg)
    assert st.tolist() == [0]
    rng = o.StdRng(v["seed"]); rng.random_fr(); rng.random_fr(); m2 = o.le32(rng.random_fq())
    st, _ = serde.verify_strings(eng, "vargen", [v["serde_signature_var_gen"]], [v["serde_public_key_var_gen"]],
                                 np.frombuffer(m2, np.uint8).reshape(1, 32))
    assert st.tolist() == [0]
    # a JSON document: the valid item, the same signature under the double scheme's first key half (wrong key),
    # and a key string whose bytes are not a point encoding (v = q): statuses 0, 2, 3
    other_pk = serde.b58encode(serde.b58decode(v["serde_public_key_double"])[32:])
    bad_pk = serde.b58encode(o.le32(o.Q))
    doc = json.dumps([{"signature": v["serde_signature"], "public_key": pk, "message": m.hex()}
                      for pk in (v["serde_public_key"], other_pk, bad_pk)])
    st, tally = serde.verify_json(eng, "single", doc)
    assert st.tolist() == [0, 2, 3] and tally.tolist() == [1, 0, 1, 1]
