"""Full-size parity by the oracle, not by construction: 2^20 items per scheme (BASELINE.json configs[1], [2], [4]
sizes), every status byte against the C restatement of the reference's algorithm on all host cores (~3.5 min of
16 threads), affine and wire entry points.  The JSON record lands in gpurun_out/soak.json (committed copies:
profiles/r02*_soak.json).  JJS_SOAK_LOG2N overrides the size (the builder's quick runs use 17)."""
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(800)
def test_soak_every_status_against_the_oracle():
    import soak_gpu
    log2n = int(os.environ.get("JJS_SOAK_LOG2N", "20"))
    rep = soak_gpu.run_soak(log2n)
    soak_gpu.write_report(rep, os.path.join(ROOT, "gpurun_out", "soak.json"))
    for scheme, r in rep["schemes"].items():
        assert r["mismatches_affine"] == 0 and r["mismatches_wire"] == 0, (scheme, r)
        assert r["gpu_tally"] == r["oracle_status_histogram"] == r["gpu_tally_wire"], (scheme, r)
        assert all(x > 0 for x in r["oracle_status_histogram"][:3]), r      # Ok, InvalidPoint, InvalidSignature all occur
    assert rep["ok"]
