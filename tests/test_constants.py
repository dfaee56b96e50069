"""Every table in the product's generated constants (csrc/jjs_constants.inc, emitted by the
self-contained jubjub_schnorr_amd/tools/gen_constants.py) against the oracle's values."""
import os
import re

import jjs_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = open(os.path.join(ROOT, "jubjub_schnorr_amd", "csrc", "jjs_constants.inc")).read()
RP = (1 << 261) % o.Q


def arrays(name):
    m = re.search(r"JJS_CONST uint32_t %s(?:\[[^\]]*\])+ = (.*?);" % name, INC, re.S)
    assert m, name
    return [[int(x, 16) for x in re.findall(r"0x([0-9a-f]+)u", grp)] for grp in re.findall(r"\{([^{}]*)\}", m.group(1))]


def limbs(v):
    assert len(v) == 9 and all(x < 1 << 29 for x in v)
    return sum(x << (29 * i) for i, x in enumerate(v))


def words(v):
    assert len(v) == 8
    return sum(x << (32 * i) for i, x in enumerate(v))


def mont(x):
    return x % o.Q * RP % o.Q


def test_field_and_curve_constants():
    q = sum(int(re.search(r"#define JJS_Q29_%d 0x([0-9a-f]+)u" % i, INC).group(1), 16) << (29 * i) for i in range(9))
    assert q == o.Q
    assert words(arrays("JJS_Q_WORDS")[0]) == o.Q
    assert words(arrays("JJS_QM2_WORDS")[0]) == o.Q - 2
    assert words(arrays("JJS_FR_WORDS")[0]) == o.R_ORDER
    assert words(arrays("JJS_FR_R2_WORDS")[0]) == pow(2, 512, o.R_ORDER)
    inv = int(re.search(r"#define JJS_FR_INV32 0x([0-9a-f]+)u", INC).group(1), 16)
    assert (inv * o.R_ORDER + 1) % (1 << 32) == 0
    assert limbs(arrays("JJS_R2")[0]) == RP * RP % o.Q
    assert limbs(arrays("JJS_ONE")[0]) == RP
    assert limbs(arrays("JJS_D")[0]) == mont(o.D)
    assert limbs(arrays("JJS_D2")[0]) == mont(2 * o.D)
    g = arrays("JJS_G"); gn = arrays("JJS_GN")
    assert (limbs(g[0]), limbs(g[1])) == (mont(o.G[0]), mont(o.G[1]))
    assert (limbs(gn[0]), limbs(gn[1])) == (mont(o.G_NUMS[0]), mont(o.G_NUMS[1]))
    assert words(arrays("JJS_DOUBLE_TAG_WORDS")[0]) == o.DOUBLE_CHALLENGE_DOMAIN


def test_poseidon_constants():
    tags = arrays("JJS_SPONGE_TAG")
    assert len(tags) == 17
    for n in range(1, 17):
        assert limbs(tags[n]) == mont(o.sponge_tag(n))
    rc = arrays("JJS_RC")
    assert len(rc) == 340
    assert [limbs(r) for r in rc] == [mont(c) for c in o.RC]
    mds = arrays("JJS_MDS")
    assert len(mds) == 25
    assert [limbs(r) for r in mds] == [mont(o.MDS[i][j]) for i in range(5) for j in range(5)]


def test_scaled_hades_tables_are_consistent():
    """The scaled small-matrix permutation is checked end to end by the Poseidon parity tests (and by the
    generator's own model against the textbook rounds); here: the matrix really is L / (i + j + 5) and the
    table shapes are what hades29.h indexes."""
    m = re.search(r"JJS_CONST uint32_t JJS_HS_MAT\[5\]\[5\] = (.*?);", INC, re.S).group(1)
    rows = [[int(x) for x in re.findall(r"\d+", grp)] for grp in re.findall(r"\{([^{}]*)\}", m)]
    assert rows == [[360360 // (i + j + 5) for j in range(5)] for i in range(5)]
    assert all(360360 % (i + j + 5) == 0 for i in range(5) for j in range(5))
    assert len(arrays("JJS_HS_RC_FULL")) == 40 and len(arrays("JJS_HS_KAPPA")) == 60 and len(arrays("JJS_HS_MU")) == 60
    # round 0 runs at scale 1: its constants are the textbook ones
    assert [limbs(r) for r in arrays("JJS_HS_RC_FULL")[:5]] == [mont(c) for c in o.RC[:5]]
    # scale recurrence: full rounds lambda' = lambda^5 * step, so after round 0 lambda_1 = step = F/L * 2^29
    step = pow(2, 256, o.Q) * pow(360360, -1, o.Q) * (1 << 29) % o.Q
    lam = 1
    for _ in range(4):
        lam = pow(lam, 5, o.Q) * step % o.Q
    assert limbs(arrays("JJS_HS_MU")[0]) == mont(pow(lam, 4, o.Q))
