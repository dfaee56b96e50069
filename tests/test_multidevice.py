"""One process driving several devices (jjs_init(k), k > 1): contiguous blocks per device, statuses
scattered back in order, tallies summed (RCCL all-reduce on real devices)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_count_argument_is_checked_without_a_gpu():
    from jubjub_schnorr_amd import _ffi
    lib = _ffi.lib()
    lib.jjs_shutdown()
    assert lib.jjs_device_count() == 0
    assert lib.jjs_init(-1) == -1 and lib.jjs_init(17) == -1
    assert b"device_count" in lib.jjs_last_error()


@pytest.mark.gpu
def test_rccl_call_sequence_on_one_device():
    import jubjub_schnorr_amd as jjs
    eng = jjs.engine()
    assert eng._lib.jjs_debug_rccl_selftest() == 0, eng._lib.jjs_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [3])
def test_host_batches_are_sharded_over_logical_devices(devices):
    # the child loads the profiling build (libjjs_gpu_prof.so): only that one can put logical devices on one card
    p = subprocess.run([sys.executable, os.path.join(HERE, "multidevice_child.py"), str(devices)],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "MULTIDEVICE OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.gpu
def test_more_devices_than_visible_is_refused():
    import torch
    import jubjub_schnorr_amd as jjs
    eng = jjs.engine()
    eng._lib.jjs_shutdown()
    try:
        assert eng._lib.jjs_init(torch.cuda.device_count() + 1) == -1
        assert b"visible" in eng._lib.jjs_last_error()
    finally:
        assert eng._lib.jjs_init(1) == 0
