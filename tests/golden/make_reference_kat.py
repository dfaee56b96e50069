#!/usr/bin/env python3
"""Extract the reference's own known-answer DATA into tests/golden/reference_kat.json.

Run in the build container only (it reads /root/reference, which does not exist on the
GPU box).  What is extracted is data -- byte arrays and base58 strings that the
reference's tests assert on -- never source text:

* src/multisig.rs:544-672  ``multisig_transcript_known_answer`` constant arrays
* tests/serde.rs:39,52,67,81,96,109,122,137  base58 vectors, all from
  ``StdRng::seed_from_u64(2321)``
* tests/common/mod.rs:23-66  the recipe constants of the legacy-double attack fixture
  (sk=17, m=23, nonce=31, R' = 37*G')
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def rust_byte_arrays(text):
    """name -> list of hex strings for every `const NAME: [[u8; N]; K]` / `[u8; N]`."""
    out = {}
    for m in re.finditer(r"const\s+([A-Z_]+):\s*(\[\[u8;\s*\d+\];\s*\d+\]|\[u8;\s*\d+\])\s*=\s*", text):
        name, ty = m.group(1), m.group(2)
        # find the matching bracketed initializer
        i = m.end()
        depth, j = 0, i
        while True:
            ch = text[j]
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        body = text[i : j + 1]
        nested = ty.startswith("[[")
        if nested:
            arrays = re.findall(r"\[([^\[\]]+)\]", body)
        else:
            arrays = [body.strip()[1:-1]]
        out[name] = ["".join("%02x" % int(t, 16) for t in re.findall(r"0x([0-9a-fA-F]{2})", a)) for a in arrays]
    return out


def main():
    ms = open(os.path.join(REF, "src/multisig.rs")).read()
    start = ms.index("fn multisig_transcript_known_answer")
    arrays = rust_byte_arrays(ms[start:])
    kat = {
        "source": "reference src/multisig.rs:544-672 (multisig_transcript_known_answer)",
        "secret_keys": [3, 5, 7],
        "r_scalars": [11, 13, 17],
        "s_scalars": [19, 23, 29],
        "message": 31,
    }
    for k, v in arrays.items():
        kat[k.lower()] = v if len(v) > 1 else v[0]

    serde = open(os.path.join(REF, "tests/serde.rs")).read()
    vec = {}
    for m in re.finditer(r"fn (serde_[a-z_]+)\(\).*?\"\\\"([1-9A-HJ-NP-Za-km-z]+)\\\"\"", serde, re.S):
        vec[m.group(1)] = m.group(2)
    want = {
        "serde_public_key", "serde_secret_key", "serde_signature", "serde_public_key_double",
        "serde_signature_double", "serde_public_key_var_gen", "serde_secret_key_var_gen",
        "serde_signature_var_gen",
    }
    assert want <= set(vec), sorted(vec)
    out = {
        "multisig_kat": kat,
        "serde_base58": {
            "source": "reference tests/serde.rs:34-142, StdRng::seed_from_u64(2321)",
            "seed": 2321,
            **{k: vec[k] for k in sorted(want)},
        },
        "legacy_double_attack": {
            "source": "reference tests/common/mod.rs:23-66 (recipe), tests/schnorr_double.rs:72-82 (expected InvalidSignature)",
            "sk": 17, "message": 23, "nonce": 31, "r_prime_scalar": 37,
            "expected_status": 2,
        },
        "double_challenge_domain_ascii": "JJSCHDBL",
    }
    with open(os.path.join(HERE, "reference_kat.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote reference_kat.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in kat.items()})


if __name__ == "__main__":
    main()
