"""The C++ host-side mirror of the reference interface (include/jjs_schnorr.hpp): compiles and links on
CPU; on the GPU it runs the reference-style scenarios from the golden vectors."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_schnorr.cpp")
PKG = os.path.join(ROOT, "jubjub_schnorr_amd")
ORDER = {"single": ["u", "R", "PK", "m"], "double": ["u", "R", "Rp", "PK", "PKp", "m"], "vargen": ["u", "R", "PK", "Gen", "m"]}


def build(tmp_path):
    exe = str(tmp_path / "test_schnorr")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L" + PKG, "-l:libjjs_gpu.so", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path):
    assert os.path.exists(os.path.join(PKG, "libjjs_gpu.so")), "run __graft_entry__.build() first"
    build(tmp_path)


@pytest.mark.gpu
def test_cpp_mirror_reference_scenarios(tmp_path):
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "verify_vectors.json")))
    lines = []
    for scheme, items in vec.items():
        for v in items:
            lines.append(" ".join([scheme, v["name"], str(v["status"])] + [v[k] for k in ORDER[scheme]]))
    # wire lines: the reference's serialised signature / key bytes (tests/serde.rs seed 2321, multisig KAT),
    # plus a corrupted encoding that from_bytes would reject
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jjs_oracle as o
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))
    v = kat["serde_base58"]
    rng = o.StdRng(v["seed"]); rng.random_fr(); m = o.le32(rng.random_fq()).hex()
    sig, pk = o.b58decode(v["serde_signature"]).hex(), o.b58decode(v["serde_public_key"]).hex()
    lines.append(f"wire single serde 0 {sig} {pk} {m}")
    k = kat["multisig_kat"]
    lines.append(f"wire single multisig_kat 0 {k['signature']} {k['aggregate_public_key']} {o.le32(k['message']).hex()}")
    lines.append(f"wire single bad_pk_encoding 3 {sig} {o.le32(o.Q).hex()} {m}")
    lines.append(f"wire single wrong_message 2 {sig} {pk} {o.le32(5).hex()}")
    lines.append(f"serde single reference_strings 0 {v['serde_signature']} {v['serde_public_key']} {m}")
    path = tmp_path / "vectors.txt"
    path.write_text("\n".join(lines) + "\n")
    exe = build(tmp_path)
    out = subprocess.run([exe, str(path)], capture_output=True, text=True)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    assert f"{len(lines)} vectors, 0 failures" in out.stdout
