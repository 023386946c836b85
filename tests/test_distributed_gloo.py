"""N>1 path on CPU: world_size 2, gloo.  Frames shard into contiguous blocks, the weight blob reaches
every rank bit-exactly through the one-time broadcast, results gather back in global frame order."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from padel_analytics_amd import dist as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 37
    blob = np.random.default_rng(0).normal(size=1000).astype(np.float32) if rank == 0 else None
    got = D.broadcast_blob(blob, 1000, src=0)
    lo, hi = D.shard_range(n, rank, world)
    local = [(i, float(got[i])) for i in range(lo, hi)]            # stand-in for per-frame results
    allr = D.gather_results(local, dst=0)
    if rank == 0:
        q.put((got[:5].tolist(), [a[0] for a in allr]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 64, 255, 256):
        for w in (1, 2, 3, 8):
            r = [D.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    head, order = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.random.default_rng(0).normal(size=1000).astype(np.float32)[:5].tolist()
    assert head == want
    assert order == list(range(37))
