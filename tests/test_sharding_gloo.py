"""The N > 1 path on CPU: world_size 2 over gloo.  Each rank verifies its contiguous shard (with the
CPU build of the product source, tests/hostbuild) and the tallies are all-reduced exactly as bench.py
does over RCCL; rank 0 checks statuses and the global tally against the oracle on the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, scheme, n, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hostlib as hl
    from helpers import ARG_ORDER, make_batch, oracle_verify
    from jubjub_schnorr_amd.sharding import shard_bounds, verify_sharded

    b = make_batch(scheme, n, seed=77, n_keys=4)
    arrays = [b[k] for k in ARG_ORDER[scheme]]

    def verify_fn(*local):
        st, tally = hl.verify(scheme, dict(zip(ARG_ORDER[scheme], local)))
        return st, torch.from_numpy(tally.astype(np.int64))

    st, (lo, hi), tally = verify_sharded(verify_fn, arrays, rank, world)
    assert (lo, hi) == shard_bounds(n, rank, world) and len(st) == hi - lo
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, st.tolist()))
    if rank == 0:
        want = oracle_verify(scheme, b)
        full = np.full(n, 255, np.uint8)
        for l, h, s in gathered:
            full[l:h] = s
        assert full.tolist() == want.tolist()
        assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scheme,n", [("single", 37), ("double", 10), ("vargen", 1)])
def test_two_rank_sharded_verify(tmp_path, scheme, n):
    sys.path.insert(0, HERE)
    import hostlib
    hostlib.load()   # build once, before forking
    mp.spawn(_worker, args=(2, _free_port(), scheme, n, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_bounds_cover_exactly():
    from jubjub_schnorr_amd.sharding import shard_bounds
    for n in (0, 1, 2, 7, 8, 9, 1 << 20, (1 << 24) + 3):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= lo <= hi <= n for lo, hi in spans)
