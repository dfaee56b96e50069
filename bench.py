#!/usr/bin/env python3
"""Benchmark: Schnorr-on-JubJub batch verification throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scheme all|single|double|vargen]
                    [--log2-items-per-gpu L] [--no-cpu-baseline] [--wire]

One "step" = one pass of the verify hot path over one batch of synthetic signatures that is
already resident in HBM.  The headline (`value`) is BASELINE.json configs[1]: 2^20 single signatures on
one MI355X; the same invocation then times configs[2] (2^20 double) and configs[4] (2^20 var-generator)
with the same protocol and reports them under `schemes`.  For N > 1 there is one process per GPU: started by
the driver (torch.distributed.run), or -- `python3 bench.py --gpus N` with WORLD_SIZE unset -- by bench.py itself as
a child process, before it has made any GPU call.  Every rank verifies its own shard -- 2^20 items per GPU at
N = 2, 4 and 2^21 per GPU at N = 8, which is configs[3] (2^24 single signatures over 8 GPUs) exactly; the N = 1
run also times 2^21 (`single_2p21`), the same-size base of that point -- and the only exchange is an RCCL
all-reduce of the 4-counter tally, inside the timed region.  `config.distributed` lists every rank's device
(ordinal, PCI bus id, uuid) and checks the all-reduced tally against the sum of the ranks' known tallies.
Rank 0 prints ONE compact JSON line; the whole record goes to gpurun_out/bench_full_n<N>.json.

Synthetic inputs: seed 0x6a6a73, 4096 distinct keys, item i signed with key i mod
4096 by the library's own GPU signer (tests pin it bit-exact to the oracle), then 15/16 valid,
1/32 wrong key, 1/64 tampered message, 1/128 identity PK, 1/256 order-2 PK, 1/256 mixed-order PK.
The expected status of every item is known by construction and is checked (bit-exact) every run.

The oracle (oracle/) is used here only as `cpu_baseline`: the C restatement of the reference's
algorithm timed on the host cores over a bounded sample of the same batch, whose statuses are also
compared with the GPU's.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x6A6A73
N_KEYS = 4096
ALGO_BYTES = {"single": 196, "double": 324, "vargen": 260}   # bytes in + status out per verify
WIRE_BYTES = {"single": 132, "double": 196, "vargen": 164}   # the same through the wire entry points
EXT_BYTES = {"single": 260, "double": 452, "vargen": 356}    # points as U || V || Z (96 B) through the _ext entry points
HBM_PEAK_GBPS = 8000.0                                       # MI355X_MICROARCH.md: 8 TB/s spec
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ARG_ORDER = {"single": ["u", "R", "PK", "m"], "double": ["u", "R", "Rp", "PK", "PKp", "m"],
             "vargen": ["u", "R", "PK", "Gen", "m"]}


def make_inputs(eng, scheme: str, n: int, rank: int, n_keys: int = N_KEYS, mix: bool = True):
    """Returns (dict of CUDA uint8 tensors, expected status tensor).  n_keys: distinct key pairs (4 096;
    n_keys = n gives every signature its own key).  mix=False: every signature valid (nothing spoilt)."""
    import torch
    gen = torch.Generator(device="cpu").manual_seed(SEED + 7919 * rank)
    N_KEYS = max(2, min(int(n_keys), n))         # noqa: N806  (shadows the module default on purpose)

    def rand_bytes(rows, top_mask):
        t = torch.randint(0, 256, (rows, 32), dtype=torch.uint8, generator=gen)
        t[:, 31] &= top_mask
        return t

    key_sk = rand_bytes(N_KEYS, 0x07); key_sk[:, 0] |= 1     # < 2^251 < r, non-zero
    key_g = rand_bytes(N_KEYS, 0x07); key_g[:, 0] |= 1
    kidx = torch.arange(n) % N_KEYS
    sk = key_sk[kidx].cuda()
    g = key_g[kidx].cuda()
    rnd = rand_bytes(n, 0x07).cuda()
    m = rand_bytes(n, 0x3F).cuda()                           # < 2^254 < q
    if scheme == "single":
        u, R, PK = eng.sign(scheme, sk, rnd, m)
        a = {"u": u, "R": R, "PK": PK, "m": m}
    elif scheme == "double":
        u, R, Rp, PK, PKp = eng.sign(scheme, sk, rnd, m)
        a = {"u": u, "R": R, "Rp": Rp, "PK": PK, "PKp": PKp, "m": m}
    else:
        u, R, PK, Gen = eng.sign(scheme, sk, rnd, m, gen_scalar=g)
        a = {"u": u, "R": R, "PK": PK, "Gen": Gen, "m": m}
    torch.cuda.synchronize()
    if not mix:
        return {k: v.contiguous() for k, v in a.items()}, torch.zeros(n, dtype=torch.uint8, device="cuda")

    sel = torch.randint(0, 256, (n,), generator=gen).cuda()
    idx = torch.arange(n, device="cuda")
    expect = torch.zeros(n, dtype=torch.uint8, device="cuda")
    # 1/32 wrong key: PK of the next key (a different key since N_KEYS > 1)
    wrong = sel < 8
    if n > 1:
        a["PK"] = torch.where(wrong[:, None], a["PK"][(idx + 1) % n], a["PK"])
        expect[wrong] = 2
    # 1/64 tampered message
    tam = (sel >= 8) & (sel < 12)
    a["m"] = a["m"].clone(); a["m"][tam, 0] ^= 1
    expect[tam] = 2
    # 1/128 identity PK, 1/256 order-2 PK, 1/256 mixed-order PK (P + (0,-1) = (-u, -v))
    ident = torch.zeros(64, dtype=torch.uint8); ident[32] = 1
    order2 = torch.tensor(list((0).to_bytes(32, "little") + (Q - 1).to_bytes(32, "little")), dtype=torch.uint8)
    idm = (sel >= 12) & (sel < 14)
    o2m = sel == 14
    mxm = sel == 15
    a["PK"][idm] = ident.cuda()
    a["PK"][o2m] = order2.cuda()
    rows = torch.nonzero(mxm).flatten()
    if len(rows):
        pk = a["PK"][rows].cpu().numpy()
        out = pk.copy()
        for j in range(len(rows)):
            uu = int.from_bytes(pk[j, :32].tobytes(), "little")
            vv = int.from_bytes(pk[j, 32:].tobytes(), "little")
            out[j, :32] = list(((Q - uu) % Q).to_bytes(32, "little"))
            out[j, 32:] = list(((Q - vv) % Q).to_bytes(32, "little"))
        a["PK"][rows] = torch.from_numpy(out).cuda()
    expect[idm | o2m | mxm] = 1
    return {k: v.contiguous() for k, v in a.items()}, expect


def host_info():
    """CPU model, cores the box has and cores this process may run on (core count stated)."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None                      # cgroup CPU bandwidth limit in cores, when the box sets one
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            a, b = parse(open(path).read())
            if b is None:
                b = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if a not in ("max", "-1"):
                quota = int(a) / int(b)
            break
        except (OSError, ValueError):
            continue
    return {"cpu_model": model, "nproc": os.cpu_count(), "affinity_cores": affinity, "cgroup_cpu_quota_cores": quota}


def pick_threads(run, info, max_threads: int) -> tuple:
    """The thread count that verifies a probe fastest, among the cores this process may use (affinity, cgroup
    quota) and halvings of it: a box that shares its sockets runs the oracle slower on 128 threads than on 16."""
    cap = max(1, min(info["affinity_cores"], max_threads))
    if info.get("cgroup_cpu_quota_cores"):
        cap = max(1, min(cap, int(info["cgroup_cpu_quota_cores"] + 0.999)))
    cands, t = [], cap
    while t >= 1 and len(cands) < 5:
        cands.append(t)
        t //= 2
    best = (0.0, 1)
    rates = {}
    for t in cands:
        r = run(t)
        rates[t] = r
        if r > best[0]:
            best = (r, t)
    return best[1], best[0], rates


def cpu_baseline(scheme: str, arrays: dict, gpu_status, gpu_challenge=None, budget_s: float = 12.0):
    """Oracle (C restatement of the reference's algorithm) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jjs_oracle_c as oc
    try:
        oc.build(native=True)
        native = True
    except Exception:
        native = False
    fn = {"single": oc.verify_single, "double": oc.verify_double, "vargen": oc.verify_vargen}[scheme]
    info = host_info()
    probe = 4096
    host = {k: arrays[k][:probe].cpu().numpy() for k in ARG_ORDER[scheme]}

    def probe_rate(t):
        t0 = time.perf_counter()
        fn(*[host[k] for k in ARG_ORDER[scheme]], threads=t, native=native)
        return probe / (time.perf_counter() - t0)

    threads, rate, probe_rates = pick_threads(probe_rate, info, oc.max_threads(native))
    n = int(min(arrays["u"].shape[0], max(probe, rate * budget_s)))
    host = {k: arrays[k][:n].cpu().numpy() for k in ARG_ORDER[scheme]}
    t0 = time.perf_counter()
    st = fn(*[host[k] for k in ARG_ORDER[scheme]], threads=threads, native=native)
    dt = time.perf_counter() - t0
    agree = bool((st == gpu_status[:n].cpu().numpy()).all())
    # BASELINE.json configs[0]: 1 024 signatures on one CPU thread
    k1 = min(n, 1024)
    t1 = time.perf_counter()
    fn(*[host[x][:k1] for x in ARG_ORDER[scheme]], threads=1, native=native)
    one_thread = k1 / (time.perf_counter() - t1)
    c_agree = None
    if gpu_challenge is not None:      # every debug challenge equal on a 2^16 sample
        k = min(n, 1 << 16)
        _, c = fn(*[host[x][:k] for x in ARG_ORDER[scheme]], threads=threads, native=native, want_c=True)
        canonical = st[:k] != 3
        c_agree = bool((c[canonical] == gpu_challenge[:k].cpu().numpy()[canonical]).all())
    return {"value": n / dt, "unit": "verifications/s", "cores": threads, "threads_used": threads, "kind": "port",
            "cpu_model": info["cpu_model"], "nproc": info["nproc"], "affinity_cores": info["affinity_cores"],
            "cgroup_cpu_quota_cores": info["cgroup_cpu_quota_cores"],
            "thread_probe": {str(k): round(v) for k, v in probe_rates.items()},
            "sample": f"first {n} items of the same {scheme} batch, oracle/jjs_oracle.c "
                      f"({'-march=native' if native else 'generic x86-64'}, OpenMP, {threads} threads), {dt:.1f} s",
            "one_thread": {"value": one_thread, "items": k1, "config": "BASELINE.json configs[0]: 1 024 signatures, CPU only"},
            "statuses_equal_gpu": agree, "challenges_equal_gpu_on_2^16_sample": c_agree}


def host_buffer_rates(eng, scheme: str, arrays: dict, expect, calls: int = 7):
    """Secondary figures, never `value`: the blocking host-buffer entry points on the same batch -- pageable numpy arrays
    in, statuses out, PCIe and the staging copies included -- for the three input formats a caller can hold: affine
    (jjs_verify_*), extended U||V||Z (jjs_verify_*_ext: what INTEGRATION.md's Rust shim passes) and wire
    (jjs_verify_*_wire: compressed points).  One untimed call, then `calls` timed ones: `value` is the median (a blocking
    call shares the host's cores with whatever else runs there; the mean and the best are reported beside it); every
    status of the last call is compared with the expectation."""
    import numpy as np
    import torch
    n = arrays["u"].shape[0]
    want = expect.cpu().numpy()
    names = ARG_ORDER[scheme]
    affine = [arrays[k].cpu().numpy() for k in names]
    zgen = torch.Generator(device="cpu").manual_seed(SEED + 99)

    def to_ext(pts):
        z = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=zgen)
        z[:, 31] &= 0x3F; z[:, 0] |= 1
        z = z.cuda()
        U = eng.debug_fq_mul(pts[:, :32].contiguous(), z)
        V = eng.debug_fq_mul(pts[:, 32:].contiguous(), z)
        return torch.cat([U, V, z], 1).contiguous().cpu().numpy()
    ext = [to_ext(arrays[k]) if arrays[k].shape[1] == 64 else arrays[k].cpu().numpy() for k in names]
    c = {k: eng.compress(v) for k, v in arrays.items() if v.shape[1] == 64}
    if scheme == "single":
        wire = [torch.cat([arrays["u"], c["R"]], 1), c["PK"], arrays["m"]]
    elif scheme == "double":
        wire = [torch.cat([arrays["u"], c["R"], c["Rp"]], 1), torch.cat([c["PK"], c["PKp"]], 1), arrays["m"]]
    else:
        wire = [torch.cat([arrays["u"], c["R"]], 1), torch.cat([c["PK"], c["Gen"]], 1), arrays["m"]]
    wire = [w.contiguous().cpu().numpy() for w in wire]
    torch.cuda.synchronize()
    out, ok = {}, True
    for fmt, fn, args, nbytes in (("affine", lambda *a: eng.verify(scheme, *a), affine, ALGO_BYTES[scheme]),
                                  ("ext", lambda *a: eng.verify_ext(scheme, *a), ext, EXT_BYTES[scheme]),
                                  ("wire", lambda *a: eng.verify_wire(scheme, *a), wire, WIRE_BYTES[scheme])):
        fn(*args)
        times = []
        for _ in range(calls):
            t0 = time.perf_counter()
            st, tally = fn(*args)
            times.append(time.perf_counter() - t0)
        good = bool((st == want).all()) and tally.tolist() == [int((want == k).sum()) for k in range(4)]
        ok = ok and good
        mean, med = sum(times) / len(times), sorted(times)[len(times) // 2]
        out[fmt] = {"value": n / med, "mean": n / mean, "best": n / min(times), "unit": "verifications/s", "ms_per_call": med * 1e3,
                    "ms_per_call_all": [round(t * 1e3, 3) for t in times], "bytes_per_item_over_pcie": nbytes, "bit_exact": good}
    out["note"] = ("blocking jjs_verify_%s{,_ext,_wire} on pageable host arrays of the same batch, PCIe inclusive, median of %d calls; "
                   "never the headline `value`" % (scheme, calls))
    return out, ok


MAX_SCLK_GHZ = 2.4          # MI355X_MICROARCH.md: peak engine clock; the ceiling when the clock cannot be sampled


def microbench_ceiling():
    """The best cycles per wave-instruction the chip issued on the mix of the Montgomery block (tools/microbench, stage
    mont-mix, priced at the clock sampled beside it): an empirical ceiling, reported next to the bound."""
    path = os.path.join(ROOT, "profiles", "microbench_r03.jsonl")
    best = None
    try:
        for line in open(path):
            if not line.startswith("{"):
                continue
            r = json.loads(line)
            if str(r.get("op", "")).startswith("mont-mix") and r.get("cycles_per_wave_instr_per_simd_at_sclk"):
                c = r["cycles_per_wave_instr_per_simd_at_sclk"]
                if r.get("waves_per_simd", 0) >= 2 and (best is None or c < best[0]):
                    best = (c, r["waves_per_simd"], r.get("sclk_mhz"))
    except (OSError, ValueError):
        return None
    return best


def alu_roofline(pmc: dict, kernel_ms: float, clocks) -> dict:
    """The binding roofline: vector-integer issue.  A SIMD of 16 lanes needs 4 cycles for a wave of 64
    lanes on the multiplier array (v_mul_lo_u32 measured at 4.00-4.08 cycles at the sampled clock, v_mad_u64_u32 at
    4.57-4.97) and 2 cycles for the simple 32-bit operations (v_add_u32, v_and_b32: 2.39-2.59): with I wave-instructions
    per launch of which I64 are of the four-cycle class (PMC: SQ_INSTS_VALU, SQ_INSTS_VALU_INT64, committed pass of THIS
    code), no schedule can need fewer than 4 I64 + 2 (I - I64) SIMD-cycles.  Against the cycles the chip had -- 1 024 SIMDs x
    the shader clock sampled beside the timed loop x the launch time -- that is `frac`, at most 1 by construction.
    Without a clock sample the peak clock stands in (a higher ceiling, a lower fraction)."""
    insts = pmc.get("valu_wave_instr_per_launch")
    if not insts:
        return None
    i64 = pmc.get("valu_int64_wave_instr_per_launch")
    share = (i64 / insts) if i64 else 1.0          # no class counts: price every instruction at four cycles (still a bound)
    floor = 4.0 * share + 2.0 * (1.0 - share)
    sclk = (clocks or {}).get("sclk_mhz_median")
    ghz = sclk / 1e3 if sclk else MAX_SCLK_GHZ
    rate = insts / (kernel_ms * 1e-3)
    peak = 1024 * ghz * 1e9 / floor
    cycles = 1024 * ghz * 1e9 / rate
    out = {"bound": "valu-issue", "achieved": rate, "peak": peak, "unit": "wave-instr/s", "frac": rate / peak,
           "sclk_ghz": ghz, "sclk_sampled": bool(sclk), "clock_samples": (clocks or {}).get("samples"),
           "power_w_median": (clocks or {}).get("power_w_median"),
           "cycles_per_wave_instr": cycles, "floor_cycles_per_wave_instr": floor, "int64_class_share": share if i64 else None,
           "valu_wave_instr_per_launch": insts, "valu_wave_instr_per_64_verifies": insts / (pmc["items"] / 64),
           "source": pmc.get("source"), "dominant_kernel": pmc.get("dominant_kernel"),
           # per kernel, from its own PMC counts and its own busy cycles (GRBM_GUI_ACTIVE), dispatches serialised: no clock needed
           "per_kernel": {k: v.get("alu_frac_alone") for k, v in (pmc.get("per_kernel") or {}).items()} or None,
           "note": "floor = 4 cycles x share of 64-bit multiply-add / shift instructions + 2 cycles x the rest; frac = floor / cycles"}
    emp = microbench_ceiling()
    if emp:
        # not a bound: the best rate the microbenchmark reached on the instruction mix of the Montgomery block, for
        # comparison (the kernels run at that rate: the ratio is 1 within the 2 % the clock moves during a stage)
        out["empirical"] = {"cycles_per_wave_instr": emp[0], "waves_per_simd": emp[1], "sclk_mhz": emp[2],
                            "ratio_to_achieved": emp[0] / cycles,
                            "source": "profiles/microbench_r03.jsonl, stage mont-mix (70 % multiply-adds on independent accumulators)"}
    return out


def csrc_hash() -> str:
    """SHA-256 over the kernel sources: ties a committed PMC profile to the code it was taken from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "jubjub_schnorr_amd", "csrc")
    for name in sorted(os.listdir(d)):
        h.update(name.encode() + b"\0")
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def committed_pmc(scheme: str, n: int):
    """Counters of the committed PMC pass for this scheme, or None when there is none for the code that is running
    (profiles/pmc_latest.json carries the csrc hash it was measured on)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get("csrc_sha256") != csrc_hash():
        return None
    rec = rec.get("schemes", {}).get(scheme)
    if not rec or rec.get("items") != n:
        return None
    return rec


def items_per_gpu(scheme: str, world: int, override) -> int:
    """BASELINE.json: configs[1]/[2]/[4] = 2^20 items on one GPU; configs[3] = 2^24 single signatures over 8 GPUs,
    i.e. 2^21 per GPU.  N = 1, 2, 4 keep 2^20 per GPU so that the N = 1 point of a scaling run equals the headline."""
    if override is not None:
        return 1 << override
    return 1 << 21 if (scheme == "single" and world == 8) else 1 << 20


def workload_name(scheme: str, n: int, world: int) -> str:
    if scheme == "single" and world == 8 and n == 1 << 21:
        return "2^24 single signatures sharded over 8 GPUs, 2^21 per GPU, resident in HBM (BASELINE.json configs[3])"
    cfg = {"single": 1, "double": 2, "vargen": 4}[scheme]
    lg = n.bit_length() - 1
    exact = " " if (n == 1 << 20) else " (size override) "
    return f"2^{lg} {scheme} signatures per GPU, resident in HBM{exact}(BASELINE.json configs[{cfg}] per GPU)"


def run_scheme(eng, scheme: str, n: int, args, dist, rank: int, world: int, with_cpu: bool, n_keys: int = N_KEYS, mix: bool = True):
    """W warm-up steps, then exactly K timed steps between two barrier + synchronize fences; returns the record of
    this scheme (rank 0) and whether every bit-exact check held (every rank)."""
    import torch
    from jubjub_schnorr_amd.sharding import allreduce_tally
    arrays, expect = make_inputs(eng, scheme, n, rank, n_keys, mix)
    call = [arrays[k] for k in ARG_ORDER[scheme]]
    if args.wire:
        c = {k: eng.compress(v) for k, v in arrays.items() if v.shape[1] == 64}
        if scheme == "single":
            wire = [torch.cat([arrays["u"], c["R"]], 1), c["PK"], arrays["m"]]
        elif scheme == "double":
            wire = [torch.cat([arrays["u"], c["R"], c["Rp"]], 1), torch.cat([c["PK"], c["PKp"]], 1), arrays["m"]]
        else:
            wire = [torch.cat([arrays["u"], c["R"]], 1), torch.cat([c["PK"], c["Gen"]], 1), arrays["m"]]
        wire = [w.contiguous() for w in wire]
        torch.cuda.synchronize()
    if args.ext:          # every point as (U, V, Z) with a per-item Z: what the Rust types hold (jjs_verify_*_ext_dev)
        zgen = torch.Generator(device="cpu").manual_seed(SEED + 1 + rank)

        def to_ext(pts):
            z = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=zgen)
            z[:, 31] &= 0x3F; z[:, 0] |= 1
            z = z.cuda()
            U = eng.debug_fq_mul(pts[:, :32].contiguous(), z)
            V = eng.debug_fq_mul(pts[:, 32:].contiguous(), z)
            return torch.cat([U, V, z], 1).contiguous()
        ext = [to_ext(arrays[k]) if arrays[k].shape[1] == 64 else arrays[k] for k in ARG_ORDER[scheme]]
        torch.cuda.synchronize()
    want_tally = torch.stack([(expect == k).sum() for k in range(4)]).to(torch.int64)

    def run_verify():
        if args.ext:
            return eng.verify_ext(scheme, *ext)
        return eng.verify_wire(scheme, *wire) if args.wire else eng.verify(scheme, *call)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        st, tally = run_verify()
        allreduce_tally(tally)              # RCCL over xGMI: 4 x int64, the path's only exchange
    fence()
    evs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        # HIP events on the stream the kernels are launched on (the engine enqueues on torch's current stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st, tally_local = run_verify()
        e1.record()
        tally = tally_local
        if dist is not None:
            tally = allreduce_tally(tally_local.clone())
        evs.append((e0, e1))
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))
    # The shader clock under this load, for the ALU roofline: sampled beside an untimed repetition of the same K steps,
    # not beside the timed ones -- the query itself (SMU round trip) costs the loop 3 % (profiles/r03_clock_sampling_ab.txt)
    clocks = None
    if rank == 0 and not args.no_clock_sampling:
        from jubjub_schnorr_amd.tools.clocks import ClockSampler
        with ClockSampler(torch.cuda.current_device()) as sampler:
            for _ in range(args.steps):
                run_verify()
            torch.cuda.synchronize()
        clocks = sampler.summary()
        if clocks:
            clocks["when"] = "an untimed repetition of the timed loop (same batches, same kernels)"

    # bit-exact check against the by-construction expectation (every rank)
    ok_status = bool(torch.equal(st, expect))
    ok_tally = bool(torch.equal(tally_local.cpu(), want_tally.cpu()))
    # the all-reduced tally against the sum of the N per-rank tallies, which are known by construction and gathered as
    # python objects (not through the collective that is being checked)
    rank_tallies = [want_tally.cpu().tolist()]
    if dist is not None:
        rank_tallies = [None] * world
        dist.all_gather_object(rank_tallies, want_tally.cpu().tolist())
    summed = [sum(t[k] for t in rank_tallies) for k in range(4)]
    ok_global = tally.cpu().tolist() == summed
    ok = ok_status and ok_tally and ok_global
    if rank != 0:
        return None, ok

    algo_bytes = WIRE_BYTES[scheme] if args.wire else (EXT_BYTES[scheme] if args.ext else ALGO_BYTES[scheme])
    achieved = algo_bytes * n / (kernel_ms * 1e-3) / 1e9
    traffic, alu = None, None
    pmc = None
    if not (args.wire or args.ext) and (n_keys == N_KEYS or n_keys >= n):
        pmc = committed_pmc(scheme + ("_unique_keys" if n_keys >= n else ""), n)
    if pmc:
        traffic = pmc.get("hbm_bytes_per_launch")
        alu = alu_roofline(pmc, kernel_ms, clocks)
    rec = {
        "value": n * world * args.steps / elapsed,
        "unit": "verifications/s",
        "ms_per_step": elapsed / args.steps * 1e3,
        "workload": workload_name(scheme, n, world),
        "items_per_gpu": n,
        "distinct_keys_per_gpu": min(n_keys, n),
        "bit_exact": {"status_vs_construction": ok_status, "tally_local": ok_tally, "tally_global": ok_global},
        "tally_allreduce": {"rank_tallies": rank_tallies, "sum_of_rank_tallies": summed, "allreduced": tally.cpu().tolist(),
                            "equal": ok_global, "ranks": len(rank_tallies)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": algo_bytes * n,
                     "kernel": ("decode_kernel + " if args.wire else "normalize_kernel + " if args.ext else "") +
                               ("key dedup / chain / table kernels + prepare_kernel + key_verify_kernel + resolve_kernel (one batch, key-table path)"
                                if min(n_keys, n) * 16 <= n and n >= 65536 else "prepare_kernel + verify_kernel + resolve_kernel (one batch)"),
                     "kernel_ms": kernel_ms,
                     "note": "integer-ALU bound path (SURVEY.md 8d): HBM is not the limiter, see alu_roofline and DESIGN.md 6; "
                             "traffic / alu_roofline are null unless profiles/pmc_latest.json was measured on this csrc hash"},
        "alu_roofline": alu,
        "clocks": clocks,
    }
    if world == 1 and not (args.wire or args.ext or args.no_two_streams) and scheme == "single" and n_keys == N_KEYS:
        # Secondary figure, never `value`: the same K batches issued on two streams in turn.  Two big calls in flight
        # sit in two call slots and fill each other's latency-bound stretches (key dedup, per-key chains, resolve pass).
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [None, None]
        for i in range(2 * 2):
            with torch.cuda.stream(streams[i & 1]):
                outs[i & 1] = run_verify()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            with torch.cuda.stream(streams[i & 1]):
                outs[i & 1] = run_verify()
        fence()
        dt = time.perf_counter() - t0
        ok2 = all(torch.equal(o[0], expect) and torch.equal(o[1].cpu(), want_tally.cpu()) for o in outs)
        rec["pipelined_two_streams"] = {"value": n * args.steps / dt, "unit": "verifications/s", "ms_per_step": dt / args.steps * 1e3,
                                        "bit_exact": bool(ok2),
                                        "note": "same K batches, issued alternately on two streams (two in flight); not the headline value"}
        ok = ok and ok2
    if world == 1 and not (args.wire or args.ext or args.no_host_buffers) and n_keys == N_KEYS:
        rec["host_buffer"], okh = host_buffer_rates(eng, scheme, arrays, expect)
        ok = ok and okh
    if with_cpu:
        gpu_c = eng.challenge(scheme, *[arrays[k][: 1 << 16] for k in ARG_ORDER[scheme][1:]])
        rec["cpu_baseline"] = cpu_baseline(scheme, arrays, st, gpu_c)
    return rec, ok


MSIG_BYTES_PER_SHARE = 32 + 3 * 64 + 1             # z, PK, R, S in; share status out
MSIG_BYTES_PER_TRANSCRIPT = 32 + 64 + 32 + 64 + 1  # m in; aggregate key, u, R, transcript status out


def run_multisig(eng, args, with_cpu: bool, log2_shares: int = 17):
    """Secondary record (never `value`): batch `verify_share` / `combine` / `aggregate_pk` (reference src/multisig.rs:154-156,
    284-387) on VALID transcripts -- tests/golden/multisig_valid_transcripts.npz (32 transcripts of 8 participants signed as
    sign_round_2 does, made by tests/golden/make_multisig_fixture.py), tiled to 2^17 shares, one share in 16 then spoilt
    (z + 1: InvalidMultisigShare for that share, no signature for its transcript).  Every output is checked against what the
    fixture holds.  cpu_baseline: the reference's algorithm (oracle/jjs_oracle.c jjo_multisig_combine) on a bounded sample."""
    import numpy as np
    import torch
    fx = np.load(os.path.join(ROOT, "tests", "golden", "multisig_valid_transcripts.npz"))
    per, base_t = int(fx["participants"]), len(fx["m"])
    base_n = per * base_t
    reps = max(1, (1 << log2_shares) // base_n)
    n, B = reps * base_n, reps * base_t
    tile = lambda a: np.tile(a, (reps, 1))                 # noqa: E731
    z = tile(fx["z"]).copy()
    rng = np.random.default_rng(SEED)
    spoil = np.zeros(n, bool)
    spoil[rng.choice(n, n // 16, replace=False)] = True
    z[spoil, 0] ^= 1                                       # another scalar < r (the top byte is untouched): the share no longer verifies
    want_share = np.where(spoil, 4, 0).astype(np.uint8)
    bad_t = spoil.reshape(B, per).any(1)
    want_t = np.where(bad_t, 4, 0).astype(np.uint8)
    want_u, want_R, want_agg = tile(fx["sig_u"]).copy(), tile(fx["sig_R"]).copy(), tile(fx["agg_pk"])
    # u = sum z_i changes with a spoilt share, but such a transcript returns no signature at all
    want_u[bad_t] = 0; want_R[bad_t] = 0
    offs = (np.arange(B + 1, dtype=np.int64) * per).astype(np.uint32)
    host = {"z": z, "PK": tile(fx["PK"]), "R": tile(fx["R"]), "S": tile(fx["S"]), "m": tile(fx["m"])}
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in host.items()}

    def call():
        return eng.multisig_combine(dev["z"], dev["PK"], dev["R"], dev["S"], dev["m"], offs)
    for _ in range(args.warmup):
        out = call()
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = call(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    st, agg, su, sr, ts = (t.cpu().numpy() for t in out)
    ok = bool((st == want_share).all() and (ts == want_t).all() and (agg == want_agg).all() and (su == want_u).all() and (sr == want_R).all())
    algo = MSIG_BYTES_PER_SHARE * n + MSIG_BYTES_PER_TRANSCRIPT * B
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    rec = {"value": n * args.steps / elapsed, "unit": "shares/s", "ms_per_step": elapsed / args.steps * 1e3, "shares": n, "transcripts": B,
           "participants_per_transcript": per, "invalid_shares": int(spoil.sum()), "transcripts_without_signature": int(bad_t.sum()),
           "bit_exact": {"share_status": bool((st == want_share).all()), "transcript_status": bool((ts == want_t).all()),
                         "aggregate_keys": bool((agg == want_agg).all()), "signatures": bool((su == want_u).all() and (sr == want_R).all())},
           "workload": "verify_share + combine + aggregate_pk over %d transcripts of %d participants (fixture tiled %d x), resident in HBM" % (B, per, reps),
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                        "traffic": None, "algorithmic_bytes_per_launch": algo, "kernel": "msig_kernel, seven passes", "kernel_ms": kernel_ms}}
    # a long transcript: the call's time is that of its longest sponge chain (timing only: random shares, statuses not
    # compared here; tests/test_gpu_parity.py checks 257 and 1 000 participants against the oracle)
    long_ms = {}
    for parts in (256, 1000):
        k = {"z": dev["z"][:parts], "PK": dev["PK"][:parts], "R": dev["R"][:parts], "S": dev["S"][:parts], "m": dev["m"][:1]}
        o1 = np.array([0, parts], np.uint32)
        eng.multisig_combine(k["z"], k["PK"], k["R"], k["S"], k["m"], o1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.multisig_combine(k["z"], k["PK"], k["R"], k["S"], k["m"], o1)
        torch.cuda.synchronize()
        long_ms[str(parts)] = round((time.perf_counter() - t1) * 1e3, 2)
    rec["one_transcript_ms"] = long_ms
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import jjs_oracle_c as oc
        try:
            oc.build(native=True)
            native = True
        except Exception:
            native = False
        info = host_info()
        probe_t = 64

        def probe_rate(t):
            t1 = time.perf_counter()
            oc.multisig_combine(*[host[k][: probe_t * per] for k in ("z", "PK", "R", "S")], host["m"][:probe_t], offs[: probe_t + 1], threads=t, native=native)
            return probe_t * per / (time.perf_counter() - t1)
        threads, rate, probe_rates = pick_threads(probe_rate, info, oc.max_threads(native))
        sample_t = int(min(B, max(probe_t, rate * 8.0 / per)))
        t1 = time.perf_counter()
        c_share, c_ts, c_agg, c_u, c_R = oc.multisig_combine(*[host[k][: sample_t * per] for k in ("z", "PK", "R", "S")], host["m"][:sample_t],
                                                              offs[: sample_t + 1], threads=threads, native=native)
        dt = time.perf_counter() - t1
        same = bool((c_share == st[: sample_t * per]).all() and (c_ts == ts[:sample_t]).all() and (c_agg == agg[:sample_t]).all() and
                    (c_u == su[:sample_t]).all() and (c_R == sr[:sample_t]).all())
        ok = ok and same
        rec["cpu_baseline"] = {"value": sample_t * per / dt, "unit": "shares/s", "cores": threads, "kind": "port", "cpu_model": info["cpu_model"],
                               "sample": "first %d transcripts (%d shares) of the same batch, oracle/jjs_oracle.c jjo_multisig_combine (the reference's "
                                         "algorithm, src/multisig.rs:326-387, 393-500), %d threads, %.1f s" % (sample_t, sample_t * per, threads, dt),
                               "outputs_equal_gpu": same}
    return rec, ok


def write_full_record(full: dict):
    """The whole record (every scheme's roofline, clocks, notes, per-call times) goes to a file; stdout carries the compact
    line, which fits the tail a driver keeps."""
    for d in (os.path.join(ROOT, "gpurun_out"), "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_full_n%d.json" % full["n_gpus"])
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        except OSError:
            continue
    return None


def _round(x, digits=4):
    return round(x, digits) if isinstance(x, float) else x


def compact_line(full: dict, path) -> dict:
    """The ONE line of stdout: the contract's keys, `roofline` with the binding (valu-issue) figures and the host-buffer
    rates of the headline scheme inside it, `cpu_baseline`, and one short record per secondary measurement."""
    def binding(alu):
        if not alu:
            return None
        return {"bound": alu["bound"], "frac": _round(alu["frac"]), "sclk_ghz": _round(alu["sclk_ghz"]), "sclk_sampled": alu["sclk_sampled"],
                "achieved": _round(alu["achieved"], 0), "peak": _round(alu["peak"], 0), "unit": alu["unit"],
                "cycles_per_wave_instr": _round(alu["cycles_per_wave_instr"]), "floor_cycles_per_wave_instr": _round(alu["floor_cycles_per_wave_instr"]),
                "valu_wave_instr_per_64_verifies": _round(alu["valu_wave_instr_per_64_verifies"], 0),
                "dominant_kernel": alu.get("dominant_kernel"), "per_kernel": alu.get("per_kernel"), "source": alu.get("source")}

    def host_rates(hb):
        if not hb:
            return None
        return {f: _round(hb[f]["value"], 0) for f in ("affine", "ext", "wire") if f in hb}

    def short(rec):
        if rec is None:
            return None
        r = rec["roofline"] if "roofline" in rec else {}
        out = {"value": _round(rec["value"], 0), "ms_per_step": _round(rec["ms_per_step"]), "bit_exact": all(rec["bit_exact"].values())}
        if r:
            out["hbm_frac"] = _round(r["frac"], 5)
            out["traffic"] = r.get("traffic")
        b = binding(rec.get("alu_roofline"))
        if b:
            out["binding_frac"] = b["frac"]; out["sclk_ghz"] = b["sclk_ghz"]
        if rec.get("host_buffer"):
            out["host_buffer"] = host_rates(rec["host_buffer"])
        if rec.get("cpu_baseline"):
            out["cpu_baseline"] = {"value": _round(rec["cpu_baseline"]["value"], 0), "cores": rec["cpu_baseline"]["cores"],
                                   "statuses_equal_gpu": rec["cpu_baseline"]["statuses_equal_gpu"]}
        return out

    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    cfg = dict(full["config"])
    d = cfg["distributed"]
    cfg["distributed"] = {"backend": d["backend"], "world_size": d["world_size"], "ranks": d["ranks"], "distinct_devices": d["distinct_devices"],
                          "rehearsal_ranks_share_devices": d["rehearsal_ranks_share_devices"],
                          "devices": [{"rank": i["rank"], "device": i["device"], "pci_bus_id": i["pci_bus_id"], "uuid": i["uuid"]} for i in d["devices"]],
                          "tally_allreduce": {k: d["tally_allreduce"][k] for k in ("sum_of_rank_tallies", "allreduced", "equal", "ranks")}}
    line["config"] = cfg
    line["bit_exact"] = full["bit_exact"]
    r = dict(full["roofline"])
    r.pop("note", None)
    b = binding(full.get("alu_roofline"))
    hb = host_rates(full.get("host_buffer"))
    # flat copies first (a record that keeps scalars only still shows them), then the objects
    if b:
        r.update({"binding_bound": b["bound"], "binding_frac": b["frac"], "binding_sclk_ghz": b["sclk_ghz"], "binding_source": b["source"]})
    if hb:
        r.update({"host_buffer_%s_per_s" % f: v for f, v in hb.items()})
    r["binding"] = b
    r["host_buffer"] = hb
    line["roofline"] = r
    if "cpu_baseline" in full:
        c = dict(full["cpu_baseline"])
        c.pop("thread_probe", None)
        line["cpu_baseline"] = c
    if "pipelined_two_streams" in full:
        line["pipelined_two_streams"] = _round(full["pipelined_two_streams"]["value"], 0)
    if "schemes" in full:
        line["schemes"] = {k: short(v) for k, v in full["schemes"].items()}
    if "unique_keys" in full:
        line["unique_keys"] = short(full["unique_keys"])
    if "single_2p21" in full:
        line["single_2p21"] = short(full["single_2p21"])
    if "single_all_valid" in full:
        line["single_all_valid"] = short(full["single_all_valid"])
    if "small_host_calls" in full:
        line["small_host_calls"] = full["small_host_calls"]
    if "multisig" in full:
        mrec = full["multisig"]
        line["multisig"] = {"value": _round(mrec["value"], 0), "unit": mrec["unit"], "ms_per_step": _round(mrec["ms_per_step"]), "shares": mrec["shares"],
                            "participants_per_transcript": mrec["participants_per_transcript"], "invalid_shares": mrec["invalid_shares"],
                            "bit_exact": all(mrec["bit_exact"].values()), "hbm_frac": _round(mrec["roofline"]["frac"], 5),
                            "one_transcript_ms": mrec["one_transcript_ms"]}
        if "cpu_baseline" in mrec:
            line["multisig"]["cpu_baseline"] = {"value": _round(mrec["cpu_baseline"]["value"], 0), "cores": mrec["cpu_baseline"]["cores"],
                                                "outputs_equal_gpu": mrec["cpu_baseline"]["outputs_equal_gpu"]}
    line["full_record"] = path
    return line


def launch_ranks(n: int, argv: list) -> int:
    """`bench.py --gpus N` without a launcher: one rank per GPU under torch.distributed.run, as a child process (this
    process has not touched the GPU).  Rank 0's JSON line is relayed on stdout, everything else on stderr."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = []
    for line in p.stdout:
        if line.startswith("{"):
            lines.append(line)
        else:
            sys.stderr.write(line)
    rc = p.wait()
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks ended without a result line\n")
        rc = 1
    return rc


def device_identity(rank: int, local_rank: int, device_index: int) -> dict:
    """What tells one GPU of a node from another: ordinal, PCI bus id, uuid (all_gather_object'ed into config.distributed)."""
    import socket
    import torch
    prop = torch.cuda.get_device_properties(device_index)
    bus = None
    if all(hasattr(prop, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bus = "%04x:%02x:%02x.0" % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id)
    uuid = getattr(prop, "uuid", None)
    return {"rank": rank, "local_rank": local_rank, "device": device_index, "pci_bus_id": bus, "uuid": str(uuid) if uuid is not None else None,
            "name": prop.name, "host": socket.gethostname(), "pid": os.getpid(),
            "visible_devices": torch.cuda.device_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scheme", default="all", choices=["all", "single", "double", "vargen", "multisig"],
                    help="all (default): headline = single, plus double and vargen in `schemes` (and the multisig batch as a secondary "
                         "record); multisig: only the multisignature batch, SURVEY.md 8(f-1), as a line of its own")
    ap.add_argument("--log2-items-per-gpu", type=int, default=None,
                    help="override the BASELINE sizes (2^20 per GPU; 2^21 for single at --gpus 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-streams", action="store_true",
                    help="skip the secondary two-stream figure (profiling passes: overlapping kernels would blur per-kernel times)")
    ap.add_argument("--no-clock-sampling", action="store_true",
                    help="do not sample the shader clock beside the timed loop (alu_roofline then prices the ceiling at the peak clock)")
    ap.add_argument("--no-host-buffers", action="store_true",
                    help="skip the secondary host-buffer figures (profiling passes)")
    ap.add_argument("--wire", action="store_true",
                    help="feed the reference's wire formats (compressed points, decoded on the device)")
    ap.add_argument("--ext", action="store_true",
                    help="feed extended coordinates (U, V, Z per point, normalised on the device)")
    ap.add_argument("--keys", type=int, default=N_KEYS,
                    help="distinct key pairs per GPU (default 4096, SURVEY.md 8d); the item count for unique keys")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the measured configuration); gloo only to rehearse the N > 1 logic on a box with "
                         "fewer GPUs than ranks (ranks then share devices; the line says so and is not a measurement)")
    ap.add_argument("--lib", default=None, help="another in-tree build of the engine (A/B timing of kernel variants)")
    args = ap.parse_args()

    # N > 1 from a plain command line (`python3 bench.py --gpus 8`): this process has made no GPU call yet (torch is not even
    # imported), so it starts the ranks as a CHILD process, relays rank 0's one JSON line and leaves with the child's code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: unset WORLD_SIZE (bench.py then starts its own ranks) "
                         f"or launch with torch.distributed.run --nproc-per-node {args.gpus}")
    rehearsal = args.backend != "nccl"
    device_index = local_rank % max(1, torch.cuda.device_count()) if rehearsal else local_rank
    torch.cuda.set_device(device_index)       # before any other GPU call of this process
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(args.backend)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        backend = dist.get_backend()
        assert dist.get_world_size() == args.gpus and backend == args.backend, (dist.get_world_size(), backend)
    identity = device_identity(rank, local_rank, device_index)
    identities = [identity]
    if dist is not None:
        identities = [None] * world
        dist.all_gather_object(identities, identity)

    if args.lib:
        from jubjub_schnorr_amd import _ffi
        _ffi.select_library(args.lib)
    import jubjub_schnorr_amd as jjs
    eng = jjs.engine()
    if args.scheme == "multisig":
        if world != 1:
            raise SystemExit("--scheme multisig is a one-GPU record")
        rec, ok = run_multisig(eng, args, not args.no_cpu_baseline)
        line = {"metric": "multisig shares verified/sec (verify_share + combine)", "value": rec["value"], "unit": rec["unit"], "n_gpus": 1,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u32 limbs (29-bit), u64 accumulators", "data": "fixture (valid transcripts), tiled",
                "config": {"workload": rec["workload"]}, "bit_exact": rec["bit_exact"], "roofline": rec["roofline"],
                "one_transcript_ms": rec["one_transcript_ms"], "invalid_shares": rec["invalid_shares"]}
        if "cpu_baseline" in rec:
            line["cpu_baseline"] = rec["cpu_baseline"]
        print(json.dumps(line), flush=True)
        if not ok:
            raise SystemExit("bit-exact check failed")
        return
    schemes = ["single", "double", "vargen"] if args.scheme == "all" else [args.scheme]
    with_cpu = (not args.no_cpu_baseline) and world == 1
    records, all_ok = {}, True
    for scheme in schemes:
        n = items_per_gpu(scheme, world, args.log2_items_per_gpu)
        rec, ok = run_scheme(eng, scheme, n, args, dist, rank, world, with_cpu, args.keys)
        records[scheme] = rec
        all_ok = all_ok and ok
        torch.cuda.empty_cache()
    extras = {}
    unique = None
    base_2p21 = None
    if args.scheme == "all" and not (args.wire or args.ext):
        # the same size with every signature under its own key: no key repeats, so the engine's key tables cannot
        # engage and every public key is a fresh variable point (the worst case for the path; round 1's number)
        n = items_per_gpu("single", world, args.log2_items_per_gpu)
        unique, ok = run_scheme(eng, "single", n, args, dist, rank, world, False, n_keys=n)
        all_ok = all_ok and ok
        torch.cuda.empty_cache()
        if world == 1 and args.log2_items_per_gpu is None:
            # the shard size of BASELINE.json configs[3] (2^24 over 8 GPUs = 2^21 per GPU) on ONE GPU: the same-size base of the
            # N = 8 point of a scaling run (N = 1, 2, 4 run 2^20 per GPU).  A secondary record, never `value`.
            lean = argparse.Namespace(**{**vars(args), "no_two_streams": True, "no_host_buffers": True, "no_clock_sampling": True})
            base_2p21, ok = run_scheme(eng, "single", 1 << 21, lean, dist, rank, world, False)
            all_ok = all_ok and ok
            torch.cuda.empty_cache()
            # configs[1] with every signature valid: what a caller whose traffic is honest sees (the mix spends 2-3 % of a
            # batch on its 1/16 of bad items: their R points' own subgroup tests run as a pass of their own behind the equations)
            all_valid, ok = run_scheme(eng, "single", 1 << 20, lean, dist, rank, world, False, mix=False)
            all_ok = all_ok and ok
            extras["single_all_valid"] = {k: all_valid[k] for k in ("value", "unit", "ms_per_step", "workload", "items_per_gpu", "bit_exact")}
            torch.cuda.empty_cache()

    if rank == 0 and world == 1 and args.scheme == "all" and not (args.wire or args.ext):
        # the multisignature batch under the same measurement contract (secondary record)
        extras["multisig"], ok = run_multisig(eng, args, with_cpu)
        all_ok = all_ok and ok
    if rank == 0 and world == 1 and args.scheme == "all" and not (args.wire or args.ext or args.no_host_buffers):
        # the reference's own call pattern: few signatures per blocking call, several host threads (secondary record)
        from jubjub_schnorr_amd.tools import small_host_calls
        try:
            extras["small_host_calls"] = small_host_calls.measure(eng, sys.modules[__name__])
            all_ok = all_ok and extras["small_host_calls"]["bit_exact"]
        except Exception as e:          # a box without gcc: the record says so, the bench goes on
            extras["small_host_calls"] = {"error": str(e)[:300]}
    if rank == 0:
        head = records[schemes[0]]
        devices = sorted({(i["host"], i["pci_bus_id"] or i["uuid"] or i["device"]) for i in identities})
        distributed = {"backend": backend, "world_size": world, "ranks": len(identities), "distinct_devices": len(devices),
                       "rehearsal_ranks_share_devices": bool(rehearsal and world > 1),
                       "devices": [{k: i[k] for k in ("rank", "local_rank", "device", "pci_bus_id", "uuid", "name", "host", "pid",
                                                      "visible_devices")} for i in identities],
                       "tally_allreduce": head["tally_allreduce"]}
        full = {
            "metric": "Schnorr verifications/sec",
            "value": head["value"],
            "unit": "verifications/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (29-bit), u64 accumulators",
            "data": "synthetic",
            "config": {"workload": head["workload"], "scheme": schemes[0], "items_per_gpu": head["items_per_gpu"],
                       "global_items": head["items_per_gpu"] * world,
                       "input_format": "wire (compressed points)" if args.wire else "extended (U, V, Z)" if args.ext else "affine",
                       "parallelism": f"batch-sharded x{world}, RCCL tally all-reduce",
                       # flat copies of what config.distributed proves (a record that keeps scalars only still shows them)
                       "backend": backend, "world_size": world, "ranks": len(identities), "distinct_devices": len(devices),
                       "device_bus_ids": ",".join(str(i["pci_bus_id"] or i["uuid"] or i["device"]) for i in identities),
                       "allreduced_tally_equals_sum_of_rank_tallies": head["tally_allreduce"]["equal"],
                       "distributed": distributed,
                       "distinct_keys_per_gpu": head["distinct_keys_per_gpu"],
                       "mix": "15/16 valid, 1/32 wrong key, 1/64 tampered m, 1/64 invalid points; 4 096 key pairs (SURVEY.md 8d)"},
            "bit_exact": head["bit_exact"],
            "roofline": head["roofline"],
            "alu_roofline": head["alu_roofline"],
            "clocks": head.get("clocks"),
        }
        for k in ("cpu_baseline", "pipelined_two_streams", "host_buffer"):
            if k in head:
                full[k] = head[k]
        if len(schemes) > 1:        # BASELINE.json metric: "single + double" (and configs[4], the per-item generator)
            full["schemes"] = {s: records[s] for s in schemes[1:]}
        if unique is not None:
            full["unique_keys"] = {k: unique[k] for k in ("value", "unit", "ms_per_step", "workload", "distinct_keys_per_gpu",
                                                           "bit_exact", "roofline", "alu_roofline", "clocks")}
            full["unique_keys"]["note"] = ("single scheme, every signature under its own public key: the key-table path cannot "
                                           "engage; `value` above is the SURVEY.md 8(d) workload, whose 4 096 keys repeat")
        if base_2p21 is not None:
            full["single_2p21"] = {k: base_2p21[k] for k in ("value", "unit", "ms_per_step", "workload", "items_per_gpu", "bit_exact")}
        for k, v in extras.items():
            full[k] = v
        path = write_full_record(full)
        print(json.dumps(compact_line(full, path)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not all_ok:
        raise SystemExit("bit-exact check failed")


if __name__ == "__main__":
    main()
