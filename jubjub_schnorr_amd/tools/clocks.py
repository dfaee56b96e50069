#!/usr/bin/env python3
"""Shader clock and board power of one GPU, sampled in-process through librocm_smi64 (a few hundred samples a second;
`rocm-smi` itself takes about a second a call).  bench.py runs a ClockSampler beside its timed loop so that the issue
ceiling of the ALU roofline is priced at the clock the chip really held; tools/microbench_clocks.py does the same around
the microbenchmark stages.  Everything here is optional: when the library or the device query is not available the
sampler reports nothing and the callers fall back to the maximum clock (a ceiling that is never too low)."""
from __future__ import annotations

import ctypes
import threading
import time

RSMI_MAX_NUM_FREQUENCIES = 33
RSMI_CLK_TYPE_SYS = 0


class _Frequencies(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * RSMI_MAX_NUM_FREQUENCIES)]


_lib = None
_lib_failed = False


def _rsmi():
    global _lib, _lib_failed
    if _lib is None and not _lib_failed:
        for name in ("librocm_smi64.so.1", "librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so"):
            try:
                lib = ctypes.CDLL(name)
                if lib.rsmi_init(ctypes.c_uint64(0)) == 0:
                    _lib = lib
                    break
            except OSError:
                continue
        if _lib is None:
            _lib_failed = True
    return _lib


def read_sclk_mhz(device: int = 0):
    """Current shader clock in MHz, or None."""
    lib = _rsmi()
    if lib is None:
        return None
    f = _Frequencies()
    if lib.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(device), ctypes.c_int(RSMI_CLK_TYPE_SYS), ctypes.byref(f)) != 0:
        return None
    if f.current >= RSMI_MAX_NUM_FREQUENCIES:
        return None
    return f.frequency[f.current] / 1e6


def read_power_w(device: int = 0):
    lib = _rsmi()
    if lib is None:
        return None
    p = ctypes.c_uint64(0)
    if lib.rsmi_dev_current_socket_power_get(ctypes.c_uint32(device), ctypes.byref(p)) != 0:
        return None
    return p.value / 1e6


class ClockSampler:
    """with ClockSampler(device) as c: <GPU work>;  then c.summary() -> {sclk_mhz_median, ..., samples} or None."""

    def __init__(self, device: int = 0, period_s: float = 0.004):
        self.device, self.period = device, period_s
        self.sclk, self.power = [], []
        self._stop = threading.Event()
        self._thread = None

    def _run(self):
        while not self._stop.is_set():
            s = read_sclk_mhz(self.device)
            if s:
                self.sclk.append(s)
            p = read_power_w(self.device)
            if p:
                self.power.append(p)
            time.sleep(self.period)

    def __enter__(self):
        if _rsmi() is not None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
        return False

    def summary(self):
        if not self.sclk:
            return None
        s = sorted(self.sclk)
        out = {"sclk_mhz_median": s[len(s) // 2], "sclk_mhz_min": s[0], "sclk_mhz_max": s[-1], "samples": len(s),
               "source": "librocm_smi64 rsmi_dev_gpu_clk_freq_get(RSMI_CLK_TYPE_SYS), sampled beside the loop"}
        if self.power:
            p = sorted(self.power)
            out["power_w_median"] = p[len(p) // 2]
        return out


if __name__ == "__main__":
    import json
    t0 = time.perf_counter()
    vals = [read_sclk_mhz(0) for _ in range(100)]
    dt = time.perf_counter() - t0
    print(json.dumps({"sclk_mhz": vals[-1], "power_w": read_power_w(0), "seconds_per_sample": dt / 100}))
