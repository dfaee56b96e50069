"""include/jjs_gpu.h as a C header: it parses as strict C99 (what cgo / bindgen / JNI generators consume), a C client
links against the library, and on the GPU that client verifies the golden vectors through the host-buffer ABI."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_client.c")
PKG = os.path.join(ROOT, "jubjub_schnorr_amd")
INC = os.path.join(ROOT, "include")
ORDER = {"single": ["u", "R", "PK", "m"], "double": ["u", "R", "Rp", "PK", "PKp", "m"], "vargen": ["u", "R", "PK", "Gen", "m"]}


def build(tmp_path):
    exe = str(tmp_path / "abi_client")
    # the profiling header only declares; its symbols are never called, so the product library satisfies the link
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-I" + INC, SRC, "-o", exe, "-L" + PKG, "-l:libjjs_gpu.so",
                           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_headers_are_strict_c99():
    for h in ("jjs_gpu.h", "jjs_gpu_profiling.h"):
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(INC, h)])


def test_c_client_compiles_and_links(tmp_path):
    assert os.path.exists(os.path.join(PKG, "libjjs_gpu.so")), "run __graft_entry__.build() first"
    build(tmp_path)


@pytest.mark.gpu
def test_c_client_verifies_the_golden_vectors(tmp_path):
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "verify_vectors.json")))
    lines = [" ".join([scheme, str(v["status"])] + [v[k] for k in ORDER[scheme]]) for scheme, items in vec.items() for v in items]
    path = tmp_path / "vectors.txt"
    path.write_text("\n".join(lines) + "\n")
    out = subprocess.run([build(tmp_path), str(path)], capture_output=True, text=True)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0 and f"{len(lines)} vectors, 0 failures" in out.stdout
