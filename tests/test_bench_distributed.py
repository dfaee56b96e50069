"""bench.py's N > 1 logic (one process per GPU, barrier + synchronize fences, MAX over ranks, tally all-reduce and the
global bit-exact check), rehearsed with two ranks that share the box's one GPU over gloo.  Not a measurement: the
driver's 8-GPU run uses RCCL and one device per rank; this only keeps that code path from rotting."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_over_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--log2-items-per-gpu", "17", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_items"] == 2 << 17 and d["scaling"] == "weak"
    assert d["config"]["distributed"] == {"backend": "gloo", "world_size": 2, "rehearsal_ranks_share_devices": True}
    for rec in [d, d["unique_keys"]] + list(d["schemes"].values()):
        assert rec["bit_exact"] == {"status_vs_construction": True, "tally_local": True, "tally_global": True}
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only


def test_world_size_must_match_the_flag():
    env = dict(os.environ, WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
