"""world_size-2 gloo test of the sharding layer (runs on CPU: the solver itself has no CPU path, so
the oracle -- the checker -- stands in for it; what is under test is daqp_amd.parallel)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daqp_amd import parallel
    from oracle import oracle as O
    n, m, ms, na, seed, _ = O.CONFIGS["C1"]
    N = 23                                   # deliberately not a multiple of the world size
    q = O.generate_batch(N, n, m, ms, na, seed)
    ora = O.Oracle()

    def stand_in(shard):
        r = ora.quadprog_batch(shard["H"], shard["f"], shard["A"], shard["bupper"], shard["blower"], None, ms=ms)
        return dict(x=r[0], lam=r[1], exitflag=r[3], iter=r[4])

    idx, res, counters = parallel.solve_sharded(stand_in, q)
    tmax = parallel.max_over_ranks(0.5 + rank)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx, x=res["x"], it=res["iter"], counters=counters, tmax=tmax)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    world, port = 2, 29517 + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    n, m, ms, na, seed, _ = O.CONFIGS["C1"]
    q = O.generate_batch(23, n, m, ms, na, seed)
    full = O.Oracle().quadprog_batch(q["H"], q["f"], q["A"], q["bupper"], q["blower"], None, ms=ms)
    seen = np.zeros(23, int)
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert np.array_equal(d["idx"], np.arange(r, 23, world))          # interleaved shards
        assert np.array_equal(d["x"], full[0][d["idx"]])                  # same answers as the unsharded batch
        seen[d["idx"]] += 1
        c = d["counters"]
        assert c.shape == (world, 3) and c[:, 0].sum() == 23              # every rank sees every rank's counters
        assert c[:, 1].sum() == full[4].sum()
        assert d["tmax"] == 1.5                                           # MAX over ranks of (0.5, 1.5)
    assert (seen == 1).all()                                              # every QP solved exactly once
