"""Child process of test_multidevice.py: one process driving several (logical) devices through the
host-buffer entry points.  Loads the profiling build of the engine (libjjs_gpu_prof.so) and switches its
logical-device mode on, so that on a one-GPU box the logical devices share the card (sharding, staging and status scatter are the real code; the tally sum is done on
the host because two ranks on one card cannot form an RCCL clique)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]

from helpers import ARG_ORDER, edge_cases, make_batch, oracle_verify  # noqa: E402
from test_gpu_parity import to_wire  # noqa: E402


def main(devices: int) -> None:
    import torch
    import jubjub_schnorr_amd as jjs
    from jubjub_schnorr_amd import _ffi
    real = torch.cuda.device_count() >= devices
    if not real:
        _ffi.select_library(_ffi.PROFILING_LIB_PATH)
        assert _ffi.lib().jjs_debug_allow_virtual_devices(1) == 0
    eng = jjs.Engine(devices)
    assert eng.device_count == devices, eng.device_count
    # which tally reduction this run exercises: with enough visible devices it must be the real one (start_comms():
    # ncclCommInitAll over the driven devices, one ncclAllReduce per host-buffer call); logical devices sharing a card
    # cannot form a clique and are summed on the host
    ranks = _ffi.lib().jjs_collective_ranks()
    print("TALLY REDUCTION:", f"RCCL clique of {ranks} ranks" if ranks else "host sum over logical devices (no clique)")
    assert ranks == (devices if real else 0), (ranks, devices, torch.cuda.device_count())
    for scheme in ("single", "double", "vargen"):
        widths = [b.shape[1] for b in (make_batch(scheme, 1)[k] for k in ARG_ORDER[scheme])]
        # n = 0, fewer items than devices (empty blocks), ragged and larger blocks
        for n in (0, 1, devices - 1, devices + 1, 1000, 5003):
            if n == 0:
                arrays = [np.zeros((0, w), np.uint8) for w in widths]
                want = np.zeros(0, np.uint8)
            else:
                b = make_batch(scheme, n, seed=900 + n, n_keys=8)
                arrays = [b[k] for k in ARG_ORDER[scheme]]
                want = oracle_verify(scheme, b)
            st, tally = eng.verify(scheme, *arrays)
            assert st.tolist() == want.tolist(), (scheme, n)
            assert tally.tolist() == [int((want == k).sum()) for k in range(4)], (scheme, n, tally)
        b = edge_cases(scheme)
        want = oracle_verify(scheme, b)
        st, tally = eng.verify(scheme, *[b[k] for k in ARG_ORDER[scheme]])
        assert st.tolist() == want.tolist() and tally.tolist() == [int((want == k).sum()) for k in range(4)]
        b = make_batch(scheme, 777, seed=31, n_keys=8)
        want = oracle_verify(scheme, b)
        st, tally = eng.verify_wire(scheme, *to_wire(scheme, b))
        assert st.tolist() == want.tolist() and tally.tolist() == [int((want == k).sum()) for k in range(4)]
    # blocks larger than one pipeline chunk (2^18 items) on every device: inputs from the device signer, every
    # 7th message tampered, statuses known by construction
    n = devices * (1 << 18) + 999
    gen = torch.Generator(device="cpu").manual_seed(4)
    def scal(top):
        t = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=gen)
        t[:, 31] &= top
        return t.cuda()
    sk, rnd, m = scal(0x07), scal(0x07), scal(0x3F)
    sk[:, 0] |= 1
    u, R, PK = eng.sign("single", sk, rnd, m)
    bad = (torch.arange(n, device="cuda") % 7) == 3
    m[bad, 0] ^= 1
    expect = (bad.to(torch.uint8) * 2).cpu().numpy()
    st, tally = eng.verify("single", *[t.cpu().numpy() for t in (u, R, PK, m)])
    assert (st == expect).all() and tally.tolist() == [int((expect == k).sum()) for k in range(4)]
    # the device-pointer calls still act on the current device
    b = make_batch("single", 300, seed=77)
    want = oracle_verify("single", b)
    st, _ = eng.verify("single", *[torch.from_numpy(b[k]).cuda() for k in ARG_ORDER["single"]])
    assert st.cpu().numpy().tolist() == want.tolist()
    print("MULTIDEVICE OK", devices)


if __name__ == "__main__":
    main(int(sys.argv[1]))
