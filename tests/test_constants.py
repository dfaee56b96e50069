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


def test_optimised_hades_tables_are_consistent():
    """The optimised permutation is checked end to end by the Poseidon parity tests; here: shapes, the
    plain MDS copy, and that the canonical-form rows are what the generator's own model uses."""
    mats = arrays("JJS_HD_MAT")
    assert len(mats) == 75
    assert [limbs(r) for r in mats[:25]] == [mont(o.MDS[i][j]) for i in range(5) for j in range(5)]
    assert len(arrays("JJS_HP_KAPPA")) == 60 and len(arrays("JJS_HP_ROWS")) == 10 and len(arrays("JJS_HF_RC")) == 40
    rows = arrays("JJS_HP_ROWS")
    assert limbs(rows[4]) == mont(1)                      # z3' = -alpha.z + 1*x
    assert limbs(rows[9]) == mont(o.MDS[4][4])            # y'  = (T^T w).z + m44*x
