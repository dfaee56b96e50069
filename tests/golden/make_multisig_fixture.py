#!/usr/bin/env python3
"""Valid multisignature transcripts for the bench's multisig record and its tests: 32 transcripts of 8 participants, every
share signed as the reference's sign_round_2 does (z = r + s*a - c*d_i*sk, /root/reference/src/multisig.rs:213-257), through
the Python oracle (oracle/jjs_oracle.py multisig_transcript, itself pinned by the reference's multisig KAT bytes).  Output:
tests/golden/multisig_valid_transcripts.npz -- data only (inputs, and what `combine` / `aggregate_pk` return for them)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import jjs_oracle as o  # noqa: E402
from helpers import fe_arr, pt_arr  # noqa: E402

TRANSCRIPTS, PARTICIPANTS, SEED = 32, 8, 0x6D756C7469


def main():
    rng = np.random.default_rng(SEED)
    rnd = lambda mod: int.from_bytes(rng.bytes(40), "little") % (mod - 1) + 1  # noqa: E731
    z, PK, R, S, m, agg, u, rsa = [], [], [], [], [], [], [], []
    for _ in range(TRANSCRIPTS):
        sks = [rnd(o.R_ORDER) for _ in range(PARTICIPANTS)]
        rs = [rnd(o.R_ORDER) for _ in range(PARTICIPANTS)]
        ss = [rnd(o.R_ORDER) for _ in range(PARTICIPANTS)]
        pks, Rs, Ss = [o.mul(o.G, x) for x in sks], [o.mul(o.G, x) for x in rs], [o.mul(o.G, x) for x in ss]
        msg = rnd(o.Q)
        ds, a_pk, a, rsa_t, c = o.multisig_transcript(pks, Rs, Ss, msg)
        zs = [(rs[i] + ss[i] * a - c * ds[i] * sks[i]) % o.R_ORDER for i in range(PARTICIPANTS)]
        (u_t, r_t), bad = o.multisig_combine(zs, pks, Rs, Ss, msg)
        assert bad is None and r_t == rsa_t
        z += zs; PK += pks; R += Rs; S += Ss; m.append(msg); agg.append(a_pk); u.append(u_t); rsa.append(rsa_t)
    np.savez_compressed(os.path.join(HERE, "multisig_valid_transcripts.npz"), participants=np.int64(PARTICIPANTS),
                        z=fe_arr(z), PK=pt_arr(PK), R=pt_arr(R), S=pt_arr(S), m=fe_arr(m), agg_pk=pt_arr(agg), sig_u=fe_arr(u), sig_R=pt_arr(rsa))


if __name__ == "__main__":
    main()
