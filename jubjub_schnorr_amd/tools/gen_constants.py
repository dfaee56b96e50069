#!/usr/bin/env python3
"""Generate jubjub_schnorr_amd/csrc/jjs_constants.inc (constants for the HIP kernels).

Self-contained on purpose: the product never imports anything under oracle/.  The
parameters are the ones SURVEY.md Appendix A pins (dusk-bls12_381 / dusk-jubjub /
dusk-poseidon as used by the reference at src/signatures.rs:130, src/signatures/double.rs:162,
src/signatures/var_gen.rs:130).  tests/test_constants.py checks every table emitted here
against the oracle.

Representation (csrc/fq29.h): a field element is 9 limbs of 29 bits ("radix 2^29"), in
Montgomery form with R' = 2^261, i.e. the stored integer is x * 2^261 mod q.  29-bit limbs
let a whole column of 32x32->64 products accumulate in one 64-bit register with no carry
handling (v_mad_u64_u32 chains), which is what the MI355X integer pipe rewards.
"""
import hashlib
import os

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R_ORDER = 0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7
LIMB_BITS, N_LIMBS = 29, 9
MONT = (1 << (LIMB_BITS * N_LIMBS)) % Q          # R' = 2^261 mod q
POSEIDON_FACTOR = (1 << 256) % Q                  # factor carried by dusk-poseidon's constants (A.3)
WIDTH, N_FULL, N_PARTIAL = 5, 8, 60
N_ROUNDS = N_FULL + N_PARTIAL
MAX_INPUTS = 16

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "csrc", "jjs_constants.inc")


def inv(a):
    return pow(a % Q, Q - 2, Q)


MASK29 = (1 << 29) - 1


def mont(x):
    return x % Q * MONT % Q


def limbs29(x):
    assert 0 <= x < 1 << (LIMB_BITS * N_LIMBS)
    return "{" + ", ".join("0x%08xu" % ((x >> (LIMB_BITS * i)) & ((1 << LIMB_BITS) - 1)) for i in range(N_LIMBS)) + "}"


def words32(x):
    return "{" + ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(8)) + "}"


D = (-(10240 * inv(10241))) % Q
G = (0x3FD2814C43AC65A6F1FBF02D0FD6CCE62E3EBB21FD6C54ED4DF7B7FFEC7BEACA, 0x12)
G_NUMS = (0x5E67B8F316F414F7BD9514C773FD4456931E316A39FE4541921710179DF76377,
          0x43D80EB3B2F3EB1B7B162DBEEB3B34FD9949BA0F82A5507A6705B707162E3EF8)
DOUBLE_TAG = int.from_bytes(b"JJSCHDBL", "big")


def round_constants():
    """c_k chain over SHA-512 of "poseidon-for-plonk"; as applied they carry a factor 2^256."""
    out, h, c = [], b"poseidon-for-plonk", 1
    for _ in range(WIDTH * N_ROUNDS):
        h = hashlib.sha512(h).digest()
        c = (int.from_bytes(h, "little") + c) % Q
        out.append(c * POSEIDON_FACTOR % Q)
    return out


def mds():
    return [[POSEIDON_FACTOR * inv(i + j + 5) % Q for j in range(WIDTH)] for i in range(WIDTH)]


def sponge_tag(n_in, n_out=1, domain=0):
    data = (0x80000000 | n_in).to_bytes(4, "big") + n_out.to_bytes(4, "big") + domain.to_bytes(8, "big")
    return int.from_bytes(hashlib.blake2b(data, digest_size=64).digest(), "little") % Q


# ---- optimised Hades --------------------------------------------------------------------------------------
# Partial round r (lanes 0..3 = p, lane 4 = y):  x = (y + c_r[4])^5 after adding c_r;  then M.  Two exact rewrites:
#  (1) the constants on lanes 0..3 commute with the S-box, so M * (c_r with lane 4 zeroed) is pushed into
#      c_{r+1}; only a scalar kappa_r on lane 4 remains per round and the last carry is folded into the
#      constants of the first full round that follows;
#  (2) with M = [[Mh, v], [w^T, m44]] the partial rounds are the time-invariant linear system
#      p' = Mh p + v x,  y' = w.p + m44 x.  In the basis T = [t0 t1 t2 t3] with t3 = v,
#      t_{i-1} = Mh t_i + alpha_i t3 (alpha = characteristic polynomial of Mh) the system is in controller
#      canonical form: z0' = z1, z1' = z2, z2' = z3, z3' = -alpha.z + x, y' = (T^T w).z + m44 x, i.e. TWO
#      five-term dot products per round and a register shift.  The change of basis is folded into the dense
#      matrices on both sides: diag(T^-1, 1) M after the last leading full round, M diag(T, 1) in the last
#      partial round.
def mat_mul(a, b):
    return [[sum(a[i][k] * b[k][j] for k in range(len(b))) % Q for j in range(len(b[0]))] for i in range(len(a))]


def mat_vec(a, x):
    return [sum(a[i][j] * x[j] for j in range(len(x))) % Q for i in range(len(a))]


def mat_inv(a):
    n = len(a)
    m = [list(r) + [int(i == j) for j in range(n)] for i, r in enumerate(a)]
    for c in range(n):
        p = next(r for r in range(c, n) if m[r][c] % Q)
        m[c], m[p] = m[p], m[c]
        iv = inv(m[c][c])
        m[c] = [x * iv % Q for x in m[c]]
        for r in range(n):
            if r != c and m[r][c]:
                f = m[r][c]
                m[r] = [(x - f * y) % Q for x, y in zip(m[r], m[c])]
    return [r[n:] for r in m]


def block_diag1(t):
    return [list(r) + [0] for r in t] + [[0, 0, 0, 0, 1]]


def optimised_hades():
    rc, m = round_constants(), mds()
    half = N_FULL // 2
    c = [rc[WIDTH * r:WIDTH * r + WIDTH] for r in range(N_ROUNDS)]
    kappa, carry = [], [0] * WIDTH
    for r in range(half, half + N_PARTIAL):
        cr = [(x + y) % Q for x, y in zip(c[r], carry)]
        kappa.append(cr[4])
        carry = mat_vec(m, cr[:4] + [0])
    full = [list(c[r]) for r in range(half)] + [list(c[r]) for r in range(half + N_PARTIAL, N_ROUNDS)]
    full[half] = [(x + y) % Q for x, y in zip(full[half], carry)]
    mh = [r[:4] for r in m[:4]]
    v = [m[i][4] for i in range(4)]
    w = m[4][:4]
    # characteristic polynomial through the Krylov vectors of v
    kry = [v]
    for _ in range(4):
        kry.append(mat_vec(mh, kry[-1]))
    kmat = [[kry[j][i] for j in range(4)] for i in range(4)]            # columns v, Mh v, Mh^2 v, Mh^3 v
    alpha = [(-x) % Q for x in mat_vec(mat_inv(kmat), kry[4])]          # Mh^4 v = -sum alpha_j Mh^j v
    t = [None, None, None, v]
    for i in (3, 2, 1):
        mv = mat_vec(mh, t[i])
        t[i - 1] = [(a + alpha[i] * b) % Q for a, b in zip(mv, v)]
    tm = [[t[j][i] for j in range(4)] for i in range(4)]                # T, columns t0..t3
    ti = mat_inv(tm)
    comp = mat_mul(mat_mul(ti, mh), tm)
    want = [[0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [(-a) % Q for a in alpha]]
    assert comp == want and mat_vec(ti, v) == [0, 0, 0, 1], "controller canonical form not reached"
    row_z3 = [(-a) % Q for a in alpha] + [1]
    row_y = [sum(w[i] * tm[i][j] for i in range(4)) % Q for j in range(4)] + [m[4][4]]
    m_pre = mat_mul(block_diag1(ti), m)
    m_last = mat_mul(m, block_diag1(tm))
    return kappa, full, row_z3, row_y, m_pre, m_last


def hades_reference(state):
    rc, m = round_constants(), mds()
    s = list(state)
    for r in range(N_ROUNDS):
        s = [(x + rc[WIDTH * r + i]) % Q for i, x in enumerate(s)]
        if r < N_FULL // 2 or r >= N_FULL // 2 + N_PARTIAL:
            s = [pow(x, 5, Q) for x in s]
        else:
            s[4] = pow(s[4], 5, Q)
        s = mat_vec(m, s)
    return s


def hades_optimised(state, kappa, full, row_z3, row_y, m_pre, m_last):
    m, s, half = mds(), list(state), N_FULL // 2
    for r in range(half):
        s = mat_vec(m_pre if r == half - 1 else m, [pow((x + full[r][i]) % Q, 5, Q) for i, x in enumerate(s)])
    for k in range(N_PARTIAL):
        x = pow((s[4] + kappa[k]) % Q, 5, Q)
        vec = s[:4] + [x]
        if k < N_PARTIAL - 1:
            s = [s[1], s[2], s[3], sum(a * b for a, b in zip(row_z3, vec)) % Q, sum(a * b for a, b in zip(row_y, vec)) % Q]
        else:
            s = mat_vec(m_last, vec)
    for r in range(half, N_FULL):
        s = mat_vec(m, [pow((x + full[r][i]) % Q, 5, Q) for i, x in enumerate(s)])
    return s


# ---- Hades with a small-integer matrix ----------------------------------------------------------------------
# The MDS matrix is Cauchy: M[i][j] = F / (i + j + 5) with F = 2^256 mod q.  With L = lcm(5..13) = 360360,
# M = (F / L) * S where S[i][j] = L / (i + j + 5) is an INTEGER below 2^17.  A row of S times the state is 5
# small-scalar x 9-limb products (45 multiply-adds) plus one Montgomery row to drop 29 bits, instead of a
# 5-term dot product with 255-bit constants (477 multiply-adds).  The scalar F/L (and the 2^-29 of the
# single Montgomery row) is never multiplied in: the state is kept as s = lambda_r * s~ with a public
# per-round scale lambda_r, which passes through the S-box as lambda^5 and is absorbed into pre-scaled
# round constants.  In partial rounds only lane 4 goes through the S-box, so it is brought back to the
# common scale by one multiplication with lambda_r^4.  After the last round the state is multiplied by
# lambda_end once.
SMALL_L = 360360
SMALL_S = [[SMALL_L // (i + j + 5) for j in range(WIDTH)] for i in range(WIDTH)]
INV_2_29 = inv(1 << 29)


def scaled_hades_constants():
    kappa, full, _, _, _, _ = optimised_hades()          # forward-pushed constants (true domain)
    f_over_l = POSEIDON_FACTOR * inv(SMALL_L) % Q
    step = f_over_l * (1 << 29) % Q                        # out_true = step * lambda' * rho
    half = N_FULL // 2
    lam = 1
    rc_full, kap, mu = [], [], []
    for r in range(N_ROUNDS):
        il = inv(lam)
        if r < half or r >= half + N_PARTIAL:
            fr = r if r < half else r - N_PARTIAL
            rc_full.append([x * il % Q for x in full[fr]])
            lam = pow(lam, 5, Q) * step % Q
        else:
            kap.append(kappa[r - half] * il % Q)
            mu.append(pow(lam, 4, Q))
            lam = lam * step % Q
    return rc_full, kap, mu, lam


def hades_scaled_model(state, rc_full, kap, mu, lam_end):
    """Exactly what the kernel does, in field arithmetic."""
    s, half, fi, pi = list(state), N_FULL // 2, 0, 0
    for r in range(N_ROUNDS):
        if r < half or r >= half + N_PARTIAL:
            t = [pow((x + rc_full[fi][i]) % Q, 5, Q) for i, x in enumerate(s)]
            fi += 1
        else:
            t = s[:4] + [pow((s[4] + kap[pi]) % Q, 5, Q) * mu[pi] % Q]
            pi += 1
        s = [sum(SMALL_S[i][j] * t[j] for j in range(WIDTH)) * INV_2_29 % Q for i in range(WIDTH)]
    return [x * lam_end % Q for x in s]


# ---- subgroup test by the order-8 Tate pairing ---------------------------------------------------
# JubJub's 2-Sylow subgroup is cyclic of order 8 (a = -1 is a square, d is not), generated by T8.
# P is in the prime-order subgroup  <=>  the reduced Tate pairing t_8(T8, P) is trivial, i.e.
#   f_{8,T8}(P)^((q-1)/8) == 1.
# Through the birational map to the Montgomery model  B y^2 = x^3 + A x^2 + x,
#   (u, v) -> (x, y) = ((1+v)/(1-v), (1+v)/((1-v) u)),
# Miller's algorithm gives f_8 = l_T^4 * l_2T^2 / (v_2T^4 * x) / B   (B: normalisation at infinity),
# where l_T, l_2T are the tangents at T8 and 2*T8 and v_2T = x - x(2*T8) = x - 1.  Clearing
# denominators modulo 8th powers, with W = (1-v) u and X = (1+v) u:
#   g = L1^4 * L2^2 * V^4 * (B * X * W)^7,   L1 = (1+v) - l1 X + c1 W,  L2 = (1+v) - l2 X,  V = X - W.
# g vanishes exactly on the 8-torsion points in the support, for which the answer is "no" as well.
T8 = (0x71D4DF38BA9E7973EAAAE086A16618D17AA41AC43DAE8582D92E6A7927200D43,
      0x4958BDB21966982E16A13035AD4D72669106EE90F384A4A1FF0D2068EFF496DD)


def ed_add(p1, p2):
    (u1, v1), (u2, v2) = p1, p2
    t = D * u1 * u2 % Q * v1 * v2 % Q
    return ((u1 * v2 + v1 * u2) * inv(1 + t) % Q, (v1 * v2 + u1 * u2) * inv(1 - t) % Q)


def sliding_window_schedule(e, width):
    """e = sum of digit * 2^pos: steps (squarings before the multiply, odd digit), then trailing squarings.
    Evaluating: acc = 1; for (n, d): acc = acc^(2^n) * g^d; finally acc = acc^(2^trailing)."""
    steps, pending, i = [], 0, e.bit_length() - 1
    while i >= 0:
        if not (e >> i) & 1:
            pending += 1
            i -= 1
            continue
        j = max(i - width + 1, 0)
        while not (e >> j) & 1:
            j += 1
        digit = (e >> j) & ((1 << (i - j + 1)) - 1)
        steps.append((pending + (i - j + 1), digit))
        pending = 0
        i = j - 1
    acc = 0
    for n, dg in steps:
        acc = (acc << n) + dg
    assert acc << pending == e and all(dg & 1 and dg < (1 << width) for _, dg in steps)
    return steps, pending


def pairing_constants():
    t2 = ed_add(T8, T8)
    t4 = ed_add(t2, t2)
    assert t4 == (0, Q - 1) and ed_add(t4, t4) == (0, 1), "T8 must have exact order 8"
    assert (T8[1] ** 2 - T8[0] ** 2 - 1 - D * T8[0] ** 2 * T8[1] ** 2) % Q == 0
    a = Q - 1
    A = 2 * (a + D) * inv(a - D) % Q
    B = 4 * inv(a - D) % Q

    def to_mont(p):
        return ((1 + p[1]) * inv(1 - p[1]) % Q, (1 + p[1]) * inv((1 - p[1]) * p[0]) % Q)

    def slope(pt):
        return (3 * pt[0] ** 2 + 2 * A * pt[0] + 1) * inv(2 * B * pt[1]) % Q

    m1, m2 = to_mont(T8), to_mont(t2)
    l1, l2 = slope(m1), slope(m2)
    assert (B * l1 * l1 - A - 2 * m1[0]) % Q == m2[0] == 1
    c1, c2 = (l1 * m1[0] - m1[1]) % Q, (l2 * m2[0] - m2[1]) % Q
    assert c2 == 0
    return {"NEG_L1": (-l1) % Q, "C1": c1, "NEG_L2": (-l2) % Q, "B": B}


def main():
    rc, m = round_constants(), mds()
    rr = (1 << 256) % R_ORDER
    L = ["// GENERATED by jubjub_schnorr_amd/tools/gen_constants.py -- do not edit.",
         "// Field elements: 9 x 29-bit limbs, Montgomery form with R' = 2^261 (see csrc/fq29.h).",
         "#define JJS_MAX_HASH_INPUTS %d" % MAX_INPUTS]
    for i in range(N_LIMBS):
        L.append("#define JJS_Q29_%d 0x%08xu" % (i, (Q >> (29 * i)) & 0x1FFFFFFF))
    L += ["JJS_CONST uint32_t JJS_Q_WORDS[8] = %s;  // q, 8 x 32-bit" % words32(Q),
          "JJS_CONST uint32_t JJS_QM2_WORDS[8] = %s;  // q - 2 (inversion exponent)" % words32(Q - 2),
          "JJS_CONST uint32_t JJS_FR_WORDS[8] = %s;  // subgroup order r, 8 x 32-bit" % words32(R_ORDER),
          "JJS_CONST uint32_t JJS_FR_R2_WORDS[8] = %s;  // 2^512 mod r" % words32(rr * rr % R_ORDER),
          "#define JJS_FR_INV32 0x%08xu  // -r^-1 mod 2^32" % ((-pow(R_ORDER, -1, 1 << 32)) % (1 << 32)),
          "JJS_CONST uint32_t JJS_R2[9] = %s;  // R'^2 mod q" % limbs29(MONT * MONT % Q),
          "JJS_CONST uint32_t JJS_ONE[9] = %s;  // R' mod q" % limbs29(MONT),
          "JJS_CONST uint32_t JJS_D[9] = %s;" % limbs29(mont(D)),
          "JJS_CONST uint32_t JJS_D2[9] = %s;  // 2d" % limbs29(mont(2 * D)),
          "JJS_CONST uint32_t JJS_G[2][9] = {%s, %s};" % (limbs29(mont(G[0])), limbs29(mont(G[1]))),
          "JJS_CONST uint32_t JJS_GN[2][9] = {%s, %s};" % (limbs29(mont(G_NUMS[0])), limbs29(mont(G_NUMS[1]))),
          "JJS_CONST uint32_t JJS_DOUBLE_TAG_WORDS[8] = %s;  // plain canonical value" % words32(DOUBLE_TAG),
          "JJS_CONST uint32_t JJS_SPONGE_TAG[JJS_MAX_HASH_INPUTS + 1][9] = {"]
    for n in range(MAX_INPUTS + 1):
        L.append("  %s," % limbs29(mont(sponge_tag(n)) if n else 0))
    L.append("};")
    pc = pairing_constants()
    L.append("// subgroup test (order-8 Tate pairing): rows {-l1, c1} for L1, then -l2 and B")
    L.append("JJS_CONST uint32_t JJS_PAIR_L1[2][9] = {%s, %s};" % (limbs29(mont(pc["NEG_L1"])), limbs29(mont(pc["C1"]))))
    L.append("JJS_CONST uint32_t JJS_PAIR_NEG_L2[9] = %s;" % limbs29(mont(pc["NEG_L2"])))
    L.append("JJS_CONST uint32_t JJS_PAIR_B[9] = %s;" % limbs29(mont(pc["B"])))
    L.append("JJS_CONST uint32_t JJS_PAIR_EXP_WORDS[8] = %s;  // (q - 1) / 8" % words32((Q - 1) // 8))
    L.append("// public exponents as left-to-right sliding-window (width 3) schedules: {squarings, odd digit} per step,")
    L.append("// then TRAILING squarings (fq_pow_schedule)")
    t_odd = (Q - 1) >> 32
    assert (Q - 1) == t_odd << 32 and t_odd & 1
    for name, e in (("PAIR", (Q - 1) // 8), ("INV", Q - 2), ("SQRT", (t_odd - 1) // 2)):
        steps, trailing = sliding_window_schedule(e, 3)
        L.append("#define JJS_%s_SW_STEPS %d  // exponent 0x%x" % (name, len(steps), e))
        L.append("#define JJS_%s_SW_TRAILING %d" % (name, trailing))
        L.append("JJS_CONST uint32_t JJS_%s_SW[JJS_%s_SW_STEPS][2] = {" % (name, name) + ", ".join("{%d, %d}" % st for st in steps) + "};")
    zeta = pow(7, t_odd, Q)
    assert pow(zeta, 1 << 31, Q) == Q - 1, "7 must be a non-residue"
    # 2-adic discrete logarithm by bytes (decode.h): bases of the seven 256-entry power tables, built on
    # the device at init, and a perfect hash of the 256 elements of the order-256 subgroup
    zi = inv(zeta)
    bases = [zi, pow(zi, 1 << 8, Q), pow(zi, 1 << 16, Q),                  # B_i[j] = zeta^(-j * 2^(8i)), i = 0..2
             zi, pow(zi, 1 << 7, Q), pow(zi, 1 << 15, Q), pow(zi, 1 << 23, Q)]  # A_0[j] = zeta^(-(j >> 1)), A_i[j] = zeta^(-j * 2^(8i-1))
    L.append("JJS_CONST uint32_t JJS_DLOG_BASES[7][9] = {" + ", ".join(limbs29(mont(x)) for x in bases) + "};")
    g8 = pow(zeta, 1 << 24, Q)
    elems = [mont(pow(g8, j, Q)) for j in range(256)]
    assert len(set(elems)) == 256
    found = None
    for mult in range(1, 1 << 12, 2):
        for shift in range(0, 14):
            keys = [(((e & 0x1FFFFFFF) * mult) >> shift) & 0xFFFF for e in elems]
            if len(set(keys)) == 256:
                found = (mult, shift)
                break
        if found:
            break
    assert found, "no collision-free hash found"
    mult, shift = found
    L.append("// dlog in the order-256 subgroup: index = ((limb0 * MULT) >> SHIFT) & 0xffff of the canonical Montgomery")
    L.append("// limbs into a 64 KiB byte table that the engine fills on the device at init (collision-free: checked here)")
    L.append("#define JJS_DLOG_HASH_MULT %du" % mult)
    L.append("#define JJS_DLOG_HASH_SHIFT %d" % shift)
    L.append("JJS_CONST uint32_t JJS_ROOT_OF_UNITY[9] = %s;  // 7^((q-1)/2^32): generator of the 2^32-torsion of Fq*" % limbs29(mont(zeta)))
    L.append("JJS_CONST uint32_t JJS_RC[%d][9] = {" % len(rc))
    L += ["  %s," % limbs29(mont(c)) for c in rc]
    L.append("};")
    kappa, full, row_z3, row_y, m_pre, m_last = optimised_hades()
    for trial in range(3):
        st = [int.from_bytes(hashlib.sha256(b"hades-selfcheck-%d-%d" % (trial, i)).digest(), "little") % Q for i in range(WIDTH)]
        assert hades_reference(st) == hades_optimised(st, kappa, full, row_z3, row_y, m_pre, m_last), "optimised Hades differs"
    rc_full, kap, mu, lam_end = scaled_hades_constants()
    assert all(SMALL_S[i][j] * (i + j + 5) == SMALL_L and SMALL_S[i][j] < 1 << 17 for i in range(5) for j in range(5))
    for trial in range(3):
        st = [int.from_bytes(hashlib.sha256(b"hades-scaled-%d-%d" % (trial, i)).digest(), "little") % Q for i in range(WIDTH)]
        assert hades_reference(st) == hades_scaled_model(st, rc_full, kap, mu, lam_end), "scaled Hades differs"
    # The S-box squares (state lane + round constant) WITHOUT a carry pass in between (fq_sqr_plus_const in
    # fq29.h): the lane has normalised limbs (< 2^29, top limb of a value below 2q), the constant is one of the
    # 100 below, and the 9-term column sums of that square must stay inside the signed 64-bit accumulator.
    top_limb = (2 * Q - 1) >> 232
    worst = 0
    for c in [x for row in rc_full for x in row] + list(kap):
        lim = [(mont(c) >> (29 * i)) & MASK29 for i in range(8)] + [mont(c) >> 232]
        x = [MASK29 + lim[i] for i in range(8)] + [top_limb + lim[8]]
        worst = max(worst, max(sum(x[i] * x[col - i] for i in range(max(0, col - 8), min(col, 8) + 1)) for col in range(17)))
    assert worst < 28 << 58, "a Hades round constant would overflow the unnormalised S-box square"
    L.append("// worst 9-term column of (lane + round constant)^2 over all round constants: %.1f * 2^58 (limit 32)" % (worst / 2 ** 58))
    L.append("// Hades with the small-integer matrix S = L / (i + j + 5), L = 360360 (scaled_hades_constants())")
    L.append("JJS_CONST uint32_t JJS_HS_MAT[5][5] = {" + ", ".join("{" + ", ".join(str(x) for x in row) + "}" for row in SMALL_S) + "};")
    assert all(SMALL_S[i][j] == SMALL_S[0][i + j] if i + j < 5 else SMALL_S[i][j] == SMALL_S[i + j - 4][4] for i in range(5) for j in range(5))
    hankel = [SMALL_S[0][k] if k < 5 else SMALL_S[k - 4][4] for k in range(9)]      # S[i][j] = hankel[i + j]
    L.append("JJS_CONST uint32_t JJS_HS_HANKEL[9] = {" + ", ".join(str(x) for x in hankel) + "};")
    L.append("JJS_CONST uint32_t JJS_HS_RC_FULL[%d][5][9] = {" % N_FULL)
    L += ["  {" + ", ".join(limbs29(mont(x)) for x in row) + "}," for row in rc_full]
    L.append("};")
    L.append("JJS_CONST uint32_t JJS_HS_KAPPA[%d][9] = {" % N_PARTIAL)
    L += ["  %s," % limbs29(mont(x)) for x in kap]
    L.append("};")
    L.append("JJS_CONST uint32_t JJS_HS_MU[%d][9] = {  // lambda_r^4: brings lane 4 back to the common scale" % N_PARTIAL)
    L += ["  %s," % limbs29(mont(x)) for x in mu]
    L.append("};")
    L.append("JJS_CONST uint32_t JJS_HS_LAMBDA_END[9] = %s;" % limbs29(mont(lam_end)))
    L.append("JJS_CONST uint32_t JJS_MDS[5][5][9] = {")
    for i in range(WIDTH):
        L.append("  {" + ", ".join(limbs29(mont(m[i][j])) for j in range(WIDTH)) + "},")
    L.append("};")
    with open(OUT, "w") as f:
        f.write("\n".join(L) + "\n")
    print("wrote", OUT)
    # SAFE tags for long transcripts (multisig: 2 + 2n and 3 + 4n inputs, n <= 256 participants): host table,
    # uploaded to the device at jjs_init
    T = ["// GENERATED by jubjub_schnorr_amd/tools/gen_constants.py -- do not edit.",
         "#define JJS_MSIG_MAX_PARTICIPANTS 256",
         "#define JJS_LONG_TAGS %d" % (3 + 4 * 256 + 1),
         "static const uint32_t JJS_SPONGE_TAG_LONG[JJS_LONG_TAGS][9] = {"]
    for n in range(3 + 4 * 256 + 1):
        T.append("  %s," % limbs29(mont(sponge_tag(n)) if n else 0))
    T.append("};")
    out2 = os.path.join(os.path.dirname(OUT), "jjs_sponge_tags_long.inc")
    with open(out2, "w") as f:
        f.write("\n".join(T) + "\n")
    print("wrote", out2)


if __name__ == "__main__":
    main()
