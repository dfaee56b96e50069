"""bench.py runs the BASELINE.json configurations it claims: shard sizes per GPU count, workload names, and the
hash that ties a committed PMC profile to the kernel sources (no GPU needed)."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_items_per_gpu_follow_the_baseline_configs():
    # configs[1]: 2^20 single on one GPU; N = 2, 4 keep 2^20 per GPU (weak scaling, N = 1 equals the headline);
    # configs[3]: 2^24 single over 8 GPUs = 2^21 per GPU
    assert [bench.items_per_gpu("single", w, None) for w in (1, 2, 4, 8)] == [1 << 20, 1 << 20, 1 << 20, 1 << 21]
    for s in ("double", "vargen"):        # configs[2], configs[4]: 2^20 per GPU at every N
        assert [bench.items_per_gpu(s, w, None) for w in (1, 2, 4, 8)] == [1 << 20] * 4
    assert bench.items_per_gpu("single", 8, 12) == 1 << 12
    assert 8 * bench.items_per_gpu("single", 8, None) == 1 << 24


def test_workload_names_cite_the_config():
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert "2^24 single" in cfgs[3] and "8 MI355X" in cfgs[3]
    assert "configs[3]" in bench.workload_name("single", 1 << 21, 8) and "2^24" in bench.workload_name("single", 1 << 21, 8)
    assert "configs[1]" in bench.workload_name("single", 1 << 20, 1)
    assert "configs[2]" in bench.workload_name("double", 1 << 20, 1)
    assert "configs[4]" in bench.workload_name("vargen", 1 << 20, 4)
    assert "override" in bench.workload_name("single", 1 << 12, 1)


def test_committed_pmc_counters_are_only_quoted_for_the_code_they_were_measured_on(tmp_path, monkeypatch):
    h = bench.csrc_hash()
    assert len(h) == 64 and h == bench.csrc_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_hash", lambda: h)
    rec = {"csrc_sha256": h, "schemes": {"single": {"items": 1 << 20, "hbm_bytes_per_launch": 1.0,
                                                    "valu_wave_instr_per_launch": 2.0, "source": "x"}}}
    (prof / "pmc_latest.json").write_text(json.dumps(rec))
    assert bench.committed_pmc("single", 1 << 20)["valu_wave_instr_per_launch"] == 2.0
    assert bench.committed_pmc("single", 1 << 21) is None and bench.committed_pmc("double", 1 << 20) is None
    rec["csrc_sha256"] = "0" * 64                      # profile of other code: nothing is quoted
    (prof / "pmc_latest.json").write_text(json.dumps(rec))
    assert bench.committed_pmc("single", 1 << 20) is None


def test_host_info_names_the_cpu():
    info = bench.host_info()
    assert info["nproc"] >= info["affinity_cores"] >= 1 and "cpu_model" in info


def test_thread_choice_prefers_the_fastest_probe():
    info = {"affinity_cores": 256, "cgroup_cpu_quota_cores": None}
    rates = {128: 17000.0, 64: 19000.0, 32: 21000.0, 16: 21700.0, 8: 11000.0}
    t, r, seen = bench.pick_threads(lambda k: rates[k], info, 128)
    assert (t, r) == (16, 21700.0) and sorted(seen) == [8, 16, 32, 64, 128]
    info = {"affinity_cores": 256, "cgroup_cpu_quota_cores": 16.0}
    t, _, seen = bench.pick_threads(lambda k: 1000.0 * k, info, 128)
    assert t == 16 and max(seen) == 16
