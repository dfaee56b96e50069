"""bench.py's N > 1 logic (one process per GPU, barrier + synchronize fences, MAX over ranks, tally all-reduce and the
global bit-exact check), rehearsed with two ranks that share the box's one GPU over gloo -- started the way the driver
starts the N = 1 bench: a plain `python3 bench.py --gpus 2`, no launcher in front.  bench.py then starts its own ranks as
a child process (before it has touched the GPU) and relays rank 0's line.  Not a measurement: the driver's 8-GPU run
uses RCCL and one device per rank; this only keeps that code path from rotting."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env_without_ranks():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_over_gloo():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--log2-items-per-gpu", "17", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=_env_without_ranks())
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                   # ONE line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_items"] == 2 << 17 and d["scaling"] == "weak"
    dist = d["config"]["distributed"]
    assert (dist["backend"], dist["world_size"], dist["ranks"]) == ("gloo", 2, 2) and dist["rehearsal_ranks_share_devices"] is True
    assert [x["rank"] for x in dist["devices"]] == [0, 1] and dist["distinct_devices"] == 1          # one card here
    # the all-reduce returned the sum of the two ranks' tallies, which are known by construction
    t = dist["tally_allreduce"]
    assert t["equal"] is True and t["ranks"] == 2 and t["allreduced"] == t["sum_of_rank_tallies"] and sum(t["allreduced"]) == 2 << 17
    assert d["config"]["ranks"] == 2 and d["config"]["allreduced_tally_equals_sum_of_rank_tallies"] is True
    assert d["bit_exact"] == {"status_vs_construction": True, "tally_local": True, "tally_global": True}
    for rec in [d["unique_keys"]] + list(d["schemes"].values()):
        assert rec["bit_exact"] is True
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only
    full = json.load(open(os.path.join(ROOT, d["full_record"])))
    assert len(full["config"]["distributed"]["tally_allreduce"]["rank_tallies"]) == 2


def test_a_set_world_size_must_match_the_flag():
    env = dict(_env_without_ranks(), WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_plain_command_starts_its_own_ranks(tmp_path, monkeypatch):
    """`bench.py --gpus N` with WORLD_SIZE unset: a child torch.distributed.run with N ranks and the same arguments, rank
    0's JSON line relayed, the child's return code returned (here the child is a stand-in that prints what it was given)."""
    import bench
    seen = {}

    class FakePopen:
        def __init__(self, cmd, stdout=None, text=None, env=None, cwd=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(["noise from a rank\n", '{"value": 1, "n_gpus": 4}\n'])

        def wait(self):
            return 0
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"])
    cmd = seen["cmd"]
    assert rc == 0 and cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the decision is taken before torch is imported: no GPU call can have happened in the parent
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(") < main.index("import torch")


def test_compact_line_keeps_the_binding_figures_inside_roofline():
    import bench
    alu = {"bound": "valu-issue", "frac": 0.861234, "sclk_ghz": 2.28, "sclk_sampled": True, "achieved": 5.6e11, "peak": 6.6e11,
           "unit": "wave-instr/s", "cycles_per_wave_instr": 4.12, "floor_cycles_per_wave_instr": 3.58,
           "valu_wave_instr_per_64_verifies": 313000.0, "source": "profiles/x.json"}
    hb = {f: {"value": v} for f, v in (("affine", 1.0e8), ("ext", 9.0e7), ("wire", 8.0e7))}
    roof = {"bound": "hbm", "achieved": 22.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0028, "traffic": 7.6e9, "kernel_ms": 9.1, "note": "long"}
    be = {"status_vs_construction": True, "tally_local": True, "tally_global": True}
    full = {"metric": "m", "value": 1.1e8, "unit": "verifications/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 9.1,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "w", "distributed": {"backend": None, "world_size": 1, "ranks": 1, "distinct_devices": 1,
                                                        "rehearsal_ranks_share_devices": False,
                                                        "devices": [{"rank": 0, "device": 0, "pci_bus_id": "0000:05:00.0", "uuid": "u"}],
                                                        "tally_allreduce": {"sum_of_rank_tallies": [1, 0, 0, 0], "allreduced": [1, 0, 0, 0],
                                                                            "equal": True, "ranks": 1, "rank_tallies": [[1, 0, 0, 0]]}}},
            "bit_exact": be, "roofline": roof, "alu_roofline": alu, "host_buffer": hb,
            "cpu_baseline": {"value": 2.0e4, "cores": 16, "thread_probe": {"16": 1}, "statuses_equal_gpu": True},
            "schemes": {"double": {"value": 6.0e7, "ms_per_step": 16.0, "bit_exact": be, "roofline": roof, "alu_roofline": alu, "host_buffer": hb}}}
    line = bench.compact_line(full, "gpurun_out/bench_full_n1.json")
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["binding"]["bound"] == "valu-issue" and r["binding_frac"] == 0.8612 and r["binding_sclk_ghz"] == 2.28
    assert r["host_buffer"] == {"affine": 1.0e8, "ext": 9.0e7, "wire": 8.0e7} and r["host_buffer_ext_per_s"] == 9.0e7
    assert line["schemes"]["double"]["binding_frac"] == 0.8612 and line["schemes"]["double"]["host_buffer"]["wire"] == 8.0e7
    assert "thread_probe" not in line["cpu_baseline"] and len(json.dumps(line)) < 6000
