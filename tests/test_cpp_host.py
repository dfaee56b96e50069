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
    path = tmp_path / "vectors.txt"
    path.write_text("\n".join(lines) + "\n")
    exe = build(tmp_path)
    out = subprocess.run([exe, str(path)], capture_output=True, text=True)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    assert f"{len(lines)} vectors, 0 failures" in out.stdout
