#!/usr/bin/env python3
"""Emit csrc/mont_asm.inc: the Montgomery product and square of fq29.h as ONE inline-asm block each, and the
Hades linear layer.

Why asm: written in C++, hipcc re-associates the column sums (each column starts from zero and the carry is
merged afterwards with an extra 64-bit add and a zero-extension move); issuing each multiply-add as its own
asm statement makes hipcc pad every dependent pair with s_nop.  One block per function avoids both: a single
dependent accumulator chain, which issues at full rate with two waves per SIMD (each wave issues a
multiply-add every ~10 cycles, profiles/microbench_r01.jsonl).

The reduction is the signed scheme of fq29.h (mont_mul_body): digits 0..7 are m_k = acc mod 2^29 and m_k*q is
SUBTRACTED (v_mad_i64_i32 against -q_j held in SGPRs), so a low column ends with one mask and one arithmetic
shift; digit 8 is taken as -(2^29 - acc mod 2^29), which adds m_8*q and keeps the result positive.
153 (117) multiply-adds + 36 masks/shifts: 189 (161) VALU instructions per product (square).

Register plan (AMDGPU function ABI: arguments v0..v17, result v0..v8, v18..v39 caller-saved):
  operands  a0..a8 read-write ("+v"): limb j of the result overwrites a_j, which is dead by column j+9
            b0..b8 inputs (product only)
  clobbers  digits in v18..v26, doubled operand of the square in v27..v34, accumulator v[36:37],
            -q_1..-q_8 in s4..s11, vcc
"""
import os

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
QL = [(Q >> (29 * i)) & 0x1FFFFFFF for i in range(9)]
assert QL[0] == 1
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "csrc", "mont_asm.inc")

ACC, ACC_LO = "v[36:37]", "v36"
M = ["v%d" % (18 + i) for i in range(9)]
D = [None] + ["v%d" % (27 + i - 1) for i in range(1, 9)]      # doubled a_1..a_8
SQ = [None] + ["s%d" % (4 + i - 1) for i in range(1, 9)]      # q_1..q_8


def prologue():
    lines = []                                  # the first multiply-add takes the literal 0 as its addend
    for i in range(1, 9):
        lines.append("s_mov_b32 %s, 0x%x" % (SQ[i], (-QL[i]) & 0xFFFFFFFF))     # -q_i as a signed 32-bit operand
    return lines


def low_epilogue(k):
    if k < 8:       # m_k = acc mod 2^29; (acc - m_k) / 2^29 is the arithmetic shift
        return ["v_and_b32 %s, 0x1fffffff, %s" % (M[k], ACC_LO), "v_ashrrev_i64 %s, 29, %s" % (ACC, ACC)]
    # last digit, kept as the negative number -(2^29 - acc mod 2^29): acc += m_8, then the shift is exact
    return ["v_and_b32 %s, 0x1fffffff, %s" % (M[8], ACC_LO),
            "v_add_u32 %s, 0xe0000000, %s" % (M[8], M[8]),
            "v_mad_i64_i32 %s, vcc, %s, -1, %s" % (ACC, M[8], ACC),
            "v_ashrrev_i64 %s, 29, %s" % (ACC, ACC)]


def high_epilogue(k, a):
    return ["v_and_b32 %s, 0x1fffffff, %s" % (a(k - 9), ACC_LO), "v_ashrrev_i64 %s, 29, %s" % (ACC, ACC)]


ACC_HI = "v37"


def mad(x, y, first=False):
    return "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, x, y, "0" if first else ACC)


def msub(m, nq):
    """acc += digit * (-q_j): digits 0..7 are non-negative (subtracts), digit 8 is negative (adds)."""
    return "v_mad_i64_i32 %s, vcc, %s, %s, %s" % (ACC, m, nq, ACC)


def body(square):
    a = lambda i: "%%[a%d]" % i  # noqa: E731
    b = lambda i: "%%[b%d]" % i  # noqa: E731
    L = prologue()
    if square:
        for i in range(1, 9):
            L.append("v_lshlrev_b32 %s, 1, %s" % (D[i], a(i)))
    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
        if square:
            for i in range(lo, hi + 1):
                if 2 * i < k:
                    L.append(mad(a(i), D[k - i]))
            if k % 2 == 0:
                L.append(mad(a(k // 2), a(k // 2), first=(k == 0)))
        else:
            for i in range(lo, hi + 1):
                L.append(mad(a(i), b(k - i), first=(k == 0)))
        for i in range(lo, hi + 1):
            if k - i >= 1 and i <= 8 and (k >= 9 or i < k):
                L.append(msub(M[i], SQ[k - i]))
        L += low_epilogue(k) if k < 9 else high_epilogue(k, a)
    # the last shift lands directly in limb 8: (acc >> 29) as one funnel shift instead of shift + move
    assert L[-1].startswith("v_ashrrev_i64")
    L[-1] = "v_alignbit_b32 %s, %s, %s, 29" % (a(8), ACC_HI, ACC_LO)
    return L


def emit(name, square):
    lines = body(square)
    n_mad = sum(l.startswith("v_mad") for l in lines) - 1          # one of them is the "+ m_8" of column 8
    assert n_mad == (117 if square else 153), n_mad
    out = ["// %d instructions, %d v_mad_u64_u32" % (len(lines), n_mad), "#define %s \\" % name]
    out += ['    "%s\\n\\t" \\' % l for l in lines]
    out[-1] = out[-1][:-2]
    return out


def hades_matrix():
    """The linear layer of one Hades round as one block: out_i = (sum_j H[i+j] * in_j + m_i * q) / 2^29 for
    i = 0..4 (fq_lincomb_small<5> five times; H = the 9 distinct entries of the Hankel matrix S, SGPR operands).
    Operands: %[o<i>_<c>] 45 early-clobber outputs, %[t<j>_<c>] 45 inputs, %[h<k>] 9 scalars; clobbers as the
    Montgomery blocks (accumulator v[36:37], quotient digit v18, q limbs s4..s11, mask s[12:13], vcc).
    One accumulator chain per row: 53 multiply-adds + 21 shifts/masks, 74 instructions (written in C++ hipcc
    restarts every column from zero and merges the carry with an extra 64-bit add: 81, plus the reloads of
    25 separately held matrix entries)."""
    L = []
    for i in range(1, 9):
        L.append("s_mov_b32 %s, 0x%x" % (SQ[i], QL[i]))
    L.append("s_mov_b64 s[12:13], 0x1fffffff")
    m = M[0]
    for i in range(5):
        for c in range(9):
            for j in range(5):
                addend = "0" if (c == 0 and j == 0) else ACC
                L.append("v_mad_u64_u32 %s, vcc, %%[t%d_%d], %%[h%d], %s" % (ACC, j, c, i + j, addend))
            if c == 0:
                L += ["v_sub_u32 %s, 0, %s" % (m, ACC_LO), "v_and_b32 %s, 0x1fffffff, %s" % (m, m),
                      "v_lshl_add_u64 %s, %s, 0, s[12:13]" % (ACC, ACC), "v_lshrrev_b64 %s, 29, %s" % (ACC, ACC)]
            else:
                L.append(mad(m, SQ[c]))      # additive digit: +m*q_c (positive q limbs here)
                L.append("v_and_b32 %%[o%d_%d], 0x1fffffff, %s" % (i, c - 1, ACC_LO))
                L.append("v_lshrrev_b64 %s, 29, %s" % (ACC, ACC) if c < 8 else
                         "v_alignbit_b32 %%[o%d_8], %s, %s, 29" % (i, ACC_HI, ACC_LO))
    n_mad = sum(l.startswith("v_mad") for l in L)
    assert n_mad == 5 * 53, n_mad
    out = ["// Hades linear layer: %d instructions, %d v_mad_u64_u32" % (len(L), n_mad), "#define JJS_HADES_MATRIX_ASM \\"]
    out += ['    "%s\\n\\t" \\' % l for l in L]
    out[-1] = out[-1][:-2]
    # operand lists for `hades_state& o` (outputs), `const fe_n (&t)[5]` (inputs) and `const uint32_t* h`
    outs = ", ".join('[o%d_%d] "=&v"(o.s[%d].l[%d])' % (i, c, i, c) for i in range(5) for c in range(9))
    ins = ", ".join('[t%d_%d] "v"(t[%d].l[%d])' % (j, c, j, c) for j in range(5) for c in range(9))
    hs = ", ".join('[h%d] "s"(h[%d])' % (k, k) for k in range(9))
    out.append("#define JJS_HADES_MATRIX_OUTPUTS " + outs)
    out.append("#define JJS_HADES_MATRIX_INPUTS " + ins + ", " + hs)
    out.append('#define JJS_HADES_MATRIX_CLOBBERS "vcc", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "v18", "v36", "v37"')
    return out


def main():
    T = ["// GENERATED by jubjub_schnorr_amd/tools/gen_mont_asm.py -- do not edit."]
    T += emit("JJS_MONT_MUL_ASM", False)
    T += emit("JJS_MONT_SQR_ASM", True)
    T += hades_matrix()
    T.append('#define JJS_MONT_ASM_CLOBBERS "vcc", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", \\')
    T.append('    "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v36", "v37"')
    with open(OUT, "w") as f:
        f.write("\n".join(T) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
