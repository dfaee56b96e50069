"""The generated Montgomery blocks of csrc/mont_asm.inc (tools/gen_mont_asm.py), interpreted instruction by instruction on the
CPU: the nine limbs they leave are a * b / 2^261 mod q (a * a for the square), below the bound fq29.h promises, for random,
extreme and weakly reduced operands.  A mistake in the generator shows up here, without a GPU.  (Round 4 also generated a
latency variant of both blocks -- the column sums gathered in a ring of eight accumulators, three dependent instructions a
column instead of one chain of 189 -- and this interpreter showed it limb-for-limb equal; on the device it was 4 % SLOWER
for the hash lanes it was meant for, profiles/r04_latency_blocks_ab.jsonl, and was removed.)"""
import os
import random
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
M64 = (1 << 64) - 1
M32 = (1 << 32) - 1


def blocks():
    text = open(os.path.join(ROOT, "jubjub_schnorr_amd", "csrc", "mont_asm.inc")).read()
    out = {}
    for name in ("JJS_MONT_MUL_ASM", "JJS_MONT_SQR_ASM"):
        body = text[text.index("#define %s \\" % name):]
        lines = []
        for line in body.splitlines()[1:]:
            m = re.match(r'\s*"(.*?)\\n\\t"', line)
            if not m:
                break
            lines.append(m.group(1))
        out[name] = lines
    return out


def s32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def s64(x):
    x &= M64
    return x - (1 << 64) if x >> 63 else x


class Machine:
    def __init__(self, a, b):
        self.r = {}
        for i in range(9):
            self.r["%%[a%d]" % i] = a[i]
            if b is not None:
                self.r["%%[b%d]" % i] = b[i]

    def get(self, op):
        op = op.strip()
        m = re.match(r"v\[(\d+):(\d+)\]$", op)
        if m:
            return self.r.get("v" + m.group(1), 0) | (self.r.get("v" + m.group(2), 0) << 32)
        if op in self.r:
            return self.r[op]
        if re.match(r"^-?(0x[0-9a-f]+|\d+)$", op):
            return int(op, 0)
        raise KeyError(op)                      # a register read before it was written

    def put(self, op, val):
        op = op.strip()
        m = re.match(r"v\[(\d+):(\d+)\]$", op)
        if m:
            self.r["v" + m.group(1)] = val & M32
            self.r["v" + m.group(2)] = (val >> 32) & M32
        else:
            self.r[op] = val & M32

    def run(self, lines):
        for line in lines:
            op, rest = line.split(None, 1)
            args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)]
            if op == "s_mov_b32":
                self.put(args[0], self.get(args[1]))
            elif op == "v_lshlrev_b32":
                self.put(args[0], (self.get(args[2]) << self.get(args[1])) & M32)
            elif op == "v_mad_u64_u32":            # dst, vcc, x, y, addend
                self.put(args[0], ((self.get(args[2]) & M32) * (self.get(args[3]) & M32) + self.get(args[4])) & M64)
            elif op == "v_mad_i64_i32":
                self.put(args[0], (s32(self.get(args[2])) * s32(self.get(args[3])) + self.get(args[4])) & M64)
            elif op == "v_lshl_add_u64":           # dst, x, shift, addend
                self.put(args[0], ((self.get(args[1]) << self.get(args[2])) + self.get(args[3])) & M64)
            elif op == "v_and_b32":
                self.put(args[0], self.get(args[1]) & self.get(args[2]))
            elif op == "v_add_u32":
                self.put(args[0], (self.get(args[1]) + self.get(args[2])) & M32)
            elif op == "v_ashrrev_i64":
                self.put(args[0], (s64(self.get(args[2])) >> self.get(args[1])) & M64)
            elif op == "v_alignbit_b32":           # dst, hi, lo, shift: (hi:lo) >> shift, low 32 bits
                self.put(args[0], (((self.get(args[1]) << 32) | self.get(args[2])) >> self.get(args[3])) & M32)
            else:
                raise ValueError(line)
        return [self.r["%%[a%d]" % i] for i in range(9)]


def limbs(x, top_bits=29):
    return [(x >> (29 * i)) & ((1 << 29) - 1) for i in range(8)] + [x >> (29 * 8)]


def value(l):
    return sum(v << (29 * i) for i, v in enumerate(l))


def test_blocks_compute_the_montgomery_product():
    B = blocks()
    assert len(B["JJS_MONT_MUL_ASM"]) == 189 + 8 and len(B["JJS_MONT_SQR_ASM"]) == 161 + 8      # + the 8 s_mov of the prologue
    rng = random.Random(29)
    rinv = pow(1 << 261, -1, Q)
    cases = [(rng.randrange(2 * Q), rng.randrange(2 * Q)) for _ in range(300)]
    cases += [(0, 0), (1, 1), (Q - 1, Q - 1), (2 * Q - 1, 2 * Q - 1), (Q, 5), ((1 << 255) - 1, (1 << 255) - 1)]
    # weakly reduced operands with limbs up to 2^30 - 1 on one side (L = 2, as fq_mul allows: La * Lb <= 3)
    for _ in range(50):
        wide = [rng.randrange(1 << 30) for _ in range(8)] + [rng.randrange(1 << 24)]
        cases.append((value(wide), rng.randrange(2 * Q)))
    for x, y in cases:
        la = limbs(x) if x < (1 << 261) else None
        lb = limbs(y)
        if x >= (1 << 261):
            continue
        # limbs may exceed 29 bits for the wide cases: rebuild them from the value only when they fit
        if any(v >> 30 for v in la):
            continue
        got = Machine(la, lb).run(B["JJS_MONT_MUL_ASM"])
        assert value(got) % Q == x * y * rinv % Q and value(got) < 2 * Q + (1 << 230)
        assert all(v < (1 << 29) for v in got[:8])
        sq = Machine(la, None).run(B["JJS_MONT_SQR_ASM"])
        assert value(sq) % Q == x * x * rinv % Q and all(v < (1 << 29) for v in sq[:8])


def test_no_register_is_read_before_it_is_written():
    """the interpreter raises KeyError on such a read"""
    B = blocks()
    a = limbs(12345678901234567890123456789 % Q)
    for name in ("JJS_MONT_MUL_ASM", "JJS_MONT_SQR_ASM"):
        Machine(a, a).run(B[name])
