"""Shared test helpers: byte packing and deterministic batch builders (oracle-side)."""
from __future__ import annotations

import numpy as np

import jjs_oracle as o
import jjs_oracle_c as oc

SEED = 0x6A6A73


def fe_bytes(x: int) -> np.ndarray:
    return np.frombuffer(o.le32(x), np.uint8).copy()


def pt_bytes(p) -> np.ndarray:
    return np.frombuffer(o.le32(p[0]) + o.le32(p[1]), np.uint8).copy()


def fe_arr(xs) -> np.ndarray:
    return np.stack([fe_bytes(x) for x in xs]) if len(xs) else np.zeros((0, 32), np.uint8)


def pt_arr(ps) -> np.ndarray:
    return np.stack([pt_bytes(p) for p in ps]) if len(ps) else np.zeros((0, 64), np.uint8)


def to_int(row) -> int:
    return int.from_bytes(np.asarray(row, np.uint8).tobytes(), "little")


def to_pt(row):
    b = np.asarray(row, np.uint8).tobytes()
    return (int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little"))


def rand_mod(rng: np.random.Generator, n: int, mod: int, nonzero=False) -> np.ndarray:
    out = np.empty((n, 32), np.uint8)
    for i in range(n):
        x = int.from_bytes(rng.bytes(48), "little") % mod
        if nonzero and x == 0:
            x = 1
        out[i] = fe_bytes(x)
    return out


ORDER2 = pt_bytes(o.ORDER2)
IDENT = pt_bytes(o.IDENTITY)


def _add_order2(points: np.ndarray) -> np.ndarray:
    return oc.point_add(points, np.tile(ORDER2, (len(points), 1)))


def torsion_generator():
    """A point of exact order 8 (the 2-Sylow subgroup of JubJub is cyclic of order 8)."""
    v = 2
    while True:
        b = bytearray(o.le32(v))
        p = o.decompress(bytes(b))
        if p is not None:
            t = o.mul(p, o.R_ORDER)
            if o.mul(t, 4) != o.IDENTITY:
                return t
        v += 1


def make_batch(scheme: str, n: int, seed: int = SEED, n_keys: int = 64, mix: bool = True):
    """Deterministic synthetic batch: K distinct keys, item i uses key i mod K,
    uniform messages, oracle signatures, then a fixed mix of corruptions.  Returns a dict of
    numpy arrays keyed by ABI argument name."""
    rng = np.random.default_rng(seed)
    K = max(1, min(n_keys, n))
    sk_k = rand_mod(rng, K, o.R_ORDER, nonzero=True)
    g_k = rand_mod(rng, K, o.R_ORDER, nonzero=True)
    idx = np.arange(n) % K
    sk = sk_k[idx]
    rnd = rand_mod(rng, n, o.R_ORDER)
    m = rand_mod(rng, n, o.Q)
    if scheme == "single":
        u, R, PK = oc.sign_single(sk, rnd, m)
        b = {"u": u, "R": R, "PK": PK, "m": m}
        pts_pk, pts_r = ["PK"], ["R"]
    elif scheme == "double":
        u, R, Rp, PK, PKp = oc.sign_double(sk, rnd, m)
        b = {"u": u, "R": R, "Rp": Rp, "PK": PK, "PKp": PKp, "m": m}
        pts_pk, pts_r = ["PK", "PKp"], ["R", "Rp"]
    elif scheme == "vargen":
        u, R, PK, Gen = oc.sign_vargen(sk, g_k[idx], rnd, m)
        b = {"u": u, "R": R, "PK": PK, "Gen": Gen, "m": m}
        pts_pk, pts_r = ["PK", "Gen"], ["R"]
    else:
        raise ValueError(scheme)
    if not mix or n == 0:
        return b
    sel = rng.integers(0, 256, size=n)
    # 15/16 valid; 1/32 wrong key; 1/64 tampered m; 1/128 identity; 1/256 order-2; 1/256 mixed order
    wrong_key = np.where(sel < 8)[0]
    tamper = np.where((sel >= 8) & (sel < 12))[0]
    ident = np.where((sel >= 12) & (sel < 14))[0]
    ord2 = np.where(sel == 14)[0]
    mixed = np.where(sel == 15)[0]
    if len(wrong_key):
        # signature made with another key -> equation fails
        other = np.roll(np.arange(n), 1)[wrong_key]
        for name in pts_pk[:1]:
            b[name][wrong_key] = b[name][other]
        # make sure "other" is really a different key
        same = (idx[wrong_key] == idx[other])
        if same.any():
            b["u"][wrong_key[same], 0] ^= 1
    if len(tamper):
        b["m"][tamper, 0] ^= 1
        # keep canonical: clear the top byte's high bits
        b["m"][tamper, 31] &= 0x3F
    for j, i in enumerate(ident):
        b[(pts_pk + pts_r)[j % len(pts_pk + pts_r)]][i] = IDENT
    for j, i in enumerate(ord2):
        b[(pts_pk + pts_r)[j % len(pts_pk + pts_r)]][i] = ORDER2
    if len(mixed):
        for j, name in enumerate(pts_pk + pts_r):
            rows = mixed[j :: len(pts_pk + pts_r)]
            if len(rows):
                b[name][rows] = _add_order2(b[name][rows])
    return b


ARG_ORDER = {
    "single": ["u", "R", "PK", "m"],
    "double": ["u", "R", "Rp", "PK", "PKp", "m"],
    "vargen": ["u", "R", "PK", "Gen", "m"],
}


def oracle_verify(scheme: str, b: dict, threads: int = 0, want_c: bool = False):
    fn = {"single": oc.verify_single, "double": oc.verify_double, "vargen": oc.verify_vargen}[scheme]
    return fn(*[b[k] for k in ARG_ORDER[scheme]], threads=threads, want_c=want_c)


def py_verify(scheme: str, b: dict, i: int) -> int:
    if scheme == "single":
        return o.verify_single(to_int(b["u"][i]), to_pt(b["R"][i]), to_pt(b["PK"][i]), to_int(b["m"][i]))
    if scheme == "double":
        return o.verify_double(to_int(b["u"][i]), to_pt(b["R"][i]), to_pt(b["Rp"][i]), to_pt(b["PK"][i]),
                               to_pt(b["PKp"][i]), to_int(b["m"][i]))
    return o.verify_vargen(to_int(b["u"][i]), to_pt(b["R"][i]), to_pt(b["PK"][i]), to_pt(b["Gen"][i]),
                           to_int(b["m"][i]))


def edge_cases(scheme: str):
    """Hand-built adversarial items, each with the status the Python oracle assigns."""
    base = make_batch(scheme, 1, seed=7, mix=False)
    items = []

    def variant(**over):
        d = {k: v[0].copy() for k, v in base.items()}
        d.update(over)
        items.append(d)

    variant()
    pts = [k for k in ARG_ORDER[scheme] if k not in ("u", "m")]
    for name in pts:
        variant(**{name: IDENT})
        variant(**{name: ORDER2})
        variant(**{name: _add_order2(base[name])[0]})
        off = base[name][0].copy(); off[32] ^= 1            # v changed -> off curve
        variant(**{name: off})
        nc = base[name][0].copy(); nc[:32] = fe_bytes(o.Q)   # u = q (non-canonical)
        variant(**{name: nc})
        nc2 = base[name][0].copy(); nc2[32:] = 0xFF          # v = 2^256-1
        variant(**{name: nc2})
    # points of order 8 and 4, and subgroup point + each torsion point
    t8 = torsion_generator()
    variant(**{pts[0]: pt_bytes(t8)})
    variant(**{pts[0]: pt_bytes(o.mul(t8, 2))})
    for k in range(1, 8):
        variant(**{pts[-1]: pt_bytes(o.add(to_pt(base[pts[-1]][0]), o.mul(t8, k)))})
    variant(u=fe_bytes(o.R_ORDER))
    variant(u=fe_bytes(o.R_ORDER - 1))
    variant(u=fe_bytes(0))
    variant(m=fe_bytes(o.Q))
    variant(m=fe_bytes(o.Q - 1))
    variant(m=fe_bytes(0))
    variant(u=np.full(32, 0xFF, np.uint8))
    b = {k: np.stack([it[k] for it in items]) for k in base}
    return b


def make_multisig_batch(n_transcripts: int, seed: int = 5, max_n: int = 5, corrupt: bool = True):
    """Random multisig transcripts through the Python oracle.  Returns arrays, offsets and, per
    transcript, the oracle's (aggregate key, a, RSa, c); expected share statuses (0 / 4)."""
    rng = np.random.default_rng(seed)
    rnd = lambda mod: int.from_bytes(rng.bytes(40), "little") % (mod - 1) + 1  # noqa: E731
    z, PK, R, S, m, offs, want, info = [], [], [], [], [], [0], [], []
    for t in range(n_transcripts):
        n = int(rng.integers(1, max_n + 1))
        sks = [rnd(o.R_ORDER) for _ in range(n)]; rs = [rnd(o.R_ORDER) for _ in range(n)]; ss = [rnd(o.R_ORDER) for _ in range(n)]
        pks = [o.mul(o.G, x) for x in sks]; Rs = [o.mul(o.G, x) for x in rs]; Ss = [o.mul(o.G, x) for x in ss]
        msg = rnd(o.Q)
        ds, agg, a, rsa, c = o.multisig_transcript(pks, Rs, Ss, msg)
        zs = [(rs[i] + ss[i] * a - c * ds[i] * sks[i]) % o.R_ORDER for i in range(n)]
        st = [0] * n
        if corrupt and t % 3 == 1:
            j = int(rng.integers(0, n)); zs[j] = (zs[j] + 1) % o.R_ORDER; st[j] = 4
        z += zs; PK += pks; R += Rs; S += Ss; m.append(msg); offs.append(offs[-1] + n); want += st
        info.append((agg, sum(zs) % o.R_ORDER, rsa))
    return (fe_arr(z), pt_arr(PK), pt_arr(R), pt_arr(S), fe_arr(m), np.array(offs, np.uint32), np.array(want, np.uint8), info)


def torsion_grid(scheme: str, seed: int = 77, extra: int = 0, reps: int = 6):
    """Signatures whose prime-order parts satisfy (or, for every third item, miss) the verification
    equation while every point carries a chosen component of the order-8 subgroup: the cases that separate a
    per-point subgroup test from any test that looks at combinations of the points.  single: all 64
    (S, T) pairs, `reps` times (the scalars' parities matter); double / vargen: all pairs on (PK, R) with
    the other points clean, then `extra` random
    assignments over all points.  Returns a batch dict like make_batch."""
    rng = np.random.default_rng(seed)
    t8 = torsion_generator()
    tors = [o.mul(t8, i) if i else o.IDENTITY for i in range(8)]
    names = {"single": ["PK", "R"], "double": ["PK", "R", "PKp", "Rp"], "vargen": ["PK", "R", "Gen"]}[scheme]
    combos = [dict(zip(names, [i, j] + [0] * (len(names) - 2))) for i in range(8) for j in range(8)] * reps
    combos += [dict.fromkeys(names, 0)] * 6          # clean points: statuses 0 and 2
    for _ in range(extra):
        combos.append({k: int(rng.integers(0, 8)) for k in names})
    rows = {k: [] for k in ARG_ORDER[scheme]}
    for idx, tc in enumerate(combos):
        sk = int.from_bytes(rng.bytes(40), "little") % (o.R_ORDER - 1) + 1
        k = int.from_bytes(rng.bytes(40), "little") % (o.R_ORDER - 1) + 1
        m = int.from_bytes(rng.bytes(40), "little") % o.Q
        wrong = int(rng.integers(0, 4) == 0)          # prime-order part of the equation off by one base point
        if scheme == "vargen":
            gen = o.add(o.mul(o.G, int.from_bytes(rng.bytes(40), "little") % (o.R_ORDER - 1) + 1), tors[tc["Gen"]])
            base = o.mul(gen, 8 * pow(8, -1, o.R_ORDER) % o.R_ORDER)      # prime-order part of Gen
        else:
            gen, base = None, o.G
        PK = o.add(o.mul(base, sk), tors[tc["PK"]])
        R = o.add(o.mul(base, k + wrong), tors[tc["R"]])
        if scheme == "single":
            c = o.challenge_single(R, PK, m)
        elif scheme == "double":
            PKp = o.add(o.mul(o.G_NUMS, sk), tors[tc["PKp"]])
            Rp = o.add(o.mul(o.G_NUMS, k + wrong), tors[tc["Rp"]])
            c = o.challenge_double(R, Rp, PK, PKp, m)
        else:
            c = o.challenge_vargen(R, PK, gen, m)
        u = (k - c * sk) % o.R_ORDER
        vals = {"u": fe_bytes(u), "R": pt_bytes(R), "PK": pt_bytes(PK), "m": fe_bytes(m)}
        if scheme == "double":
            vals.update(Rp=pt_bytes(Rp), PKp=pt_bytes(PKp))
        if scheme == "vargen":
            vals.update(Gen=pt_bytes(gen))
        for key in rows:
            rows[key].append(vals[key])
    return {key: np.stack(v) for key, v in rows.items()}


def to_extended(points: np.ndarray, rng: np.random.Generator, z_one_every: int = 7) -> np.ndarray:
    """Affine (n, 64) -> extended (n, 96) = U || V || Z with a random non-zero Z per point (Z = 1 for every
    `z_one_every`-th): the same projective point the Rust type could hold after any chain of additions."""
    out = np.empty((len(points), 96), np.uint8)
    for i, row in enumerate(points):
        u, v = to_pt(row)
        z = 1 if (z_one_every and i % z_one_every == 0) else int.from_bytes(rng.bytes(40), "little") % (o.Q - 1) + 1
        out[i, :32] = fe_bytes(u * z % o.Q); out[i, 32:64] = fe_bytes(v * z % o.Q); out[i, 64:] = fe_bytes(z)
    return out


def ext_on_device(eng, points: np.ndarray, seed: int = 11) -> np.ndarray:
    """to_extended for large arrays: U = u*Z, V = v*Z through the engine's field multiplier (itself pinned against the
    oracle by test_fq_mul).  Rows with a non-canonical coordinate cannot be rescaled: they keep their bytes with Z = 1."""
    import torch
    n = len(points)
    z = np.random.default_rng(seed).integers(0, 256, (n, 32), dtype=np.uint8)
    z[:, 31] &= 0x3F; z[:, 0] |= 1
    big = np.zeros(n, bool)
    for half in (points[:, :32], points[:, 32:]):
        for i in np.where(half[:, 31] >= 0x73)[0]:
            big[i] = big[i] or to_int(half[i]) >= o.Q
    z[big] = fe_bytes(1)
    zd = torch.from_numpy(z).cuda()
    cd = torch.from_numpy(np.ascontiguousarray(points)).cuda()
    U = eng.debug_fq_mul(cd[:, :32].contiguous(), zd).cpu().numpy()
    V = eng.debug_fq_mul(cd[:, 32:].contiguous(), zd).cpu().numpy()
    out = np.concatenate([U, V, z], 1)
    out[big, :64] = points[big]
    return np.ascontiguousarray(out)


def batch_to_extended(scheme: str, b: dict, seed: int = 3) -> list:
    """The arrays of a batch in ABI order with every point array turned into extended coordinates."""
    rng = np.random.default_rng(seed)
    return [to_extended(b[k], rng) if b[k].shape[1] == 64 else b[k] for k in ARG_ORDER[scheme]]


def crafted_collision_batch(n_good=1 << 17, n_bad=2000):
    """A single-signature batch whose last n_bad public keys are distinct byte strings built to land on ONE slot of the
    engine's key hash table (csrc/key_tables.h: the hash is public and unkeyed, so a sender can do this)."""
    M64 = (1 << 64) - 1
    C = 0xFF51AFD7ED558CCD
    C_INV = pow(C, -1, 1 << 64)

    def state_before_last_chunk(key56: bytes) -> int:
        h = 0x9E3779B97F4A7C15
        for k in range(7):
            h ^= int.from_bytes(key56[8 * k:8 * k + 8], "little")
            h = (h * C) & M64
            h ^= h >> 29
        return h

    n = n_good + n_bad
    slots = 1
    while slots < 2 * n:
        slots <<= 1
    mask = slots - 1
    b = make_batch("single", n, seed=31337, n_keys=512)
    rng = np.random.default_rng(5)
    target = 0x2A5A5 & mask
    for i in range(n_good, n):
        prefix = rng.bytes(56)
        t_hi = int.from_bytes(rng.bytes(8), "little") & ~mask & M64
        t = t_hi | (target ^ ((t_hi >> 29) & mask))
        x = ((t * C_INV) & M64) ^ state_before_last_chunk(prefix)
        key = prefix + x.to_bytes(8, "little")
        # the engine's hash of these 64 bytes ends on `target`
        h = state_before_last_chunk(prefix) ^ x
        h = (h * C) & M64
        h ^= h >> 29
        assert (h & 0xFFFFFFFF) & mask == target
        b["PK"][i] = np.frombuffer(key, np.uint8)
    return b


def to_wire(scheme, b):
    """The reference's byte formats of a batch: (signature bytes, key bytes, messages)."""
    comp = lambda a: np.stack([np.frombuffer(o.compress((int.from_bytes(r[:32].tobytes(), "little"),  # noqa: E731
                                                         int.from_bytes(r[32:].tobytes(), "little"))), np.uint8) for r in a])
    if scheme == "single":
        return np.concatenate([b["u"], comp(b["R"])], 1), comp(b["PK"]), b["m"]
    if scheme == "double":
        return np.concatenate([b["u"], comp(b["R"]), comp(b["Rp"])], 1), np.concatenate([comp(b["PK"]), comp(b["PKp"])], 1), b["m"]
    return np.concatenate([b["u"], comp(b["R"])], 1), np.concatenate([comp(b["PK"]), comp(b["Gen"])], 1), b["m"]
