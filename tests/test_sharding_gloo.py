"""N > 1 path on CPU: two gloo ranks.  Checks what bench.py relies on: each rank owns its own slice
of the seeded instance stream (disjoint, and the concatenation equals the single-process stream),
the barrier works, and the elapsed time is reduced with MAX over ranks."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_dir, cfg=2):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from asif_amd import dist, workloads
    g = dist.Group(backend="gloo")
    first, count = g.shard(B)
    x, u = workloads.make_batch(cfg, count, first=first)
    g.barrier()
    tmax = g.max_over_ranks(1.0 + rank)  # rank 1 is "slower"
    total = g.sum_over_ranks(count)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), x=x, u=u, first=first, tmax=tmax, total=total)
    g.close()


def test_two_rank_shards(tmp_path):
    B, world = 1000, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    from asif_amd import workloads
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    assert [int(v["first"]) for v in r] == [0, B]
    whole_x, whole_u = workloads.make_batch(2, world * B)
    assert np.array_equal(np.concatenate([v["x"] for v in r], axis=1), whole_x)
    assert np.array_equal(np.concatenate([v["u"] for v in r], axis=1), whole_u)
    assert all(float(v["tmax"]) == 2.0 for v in r)        # MAX over ranks, seen by every rank
    assert all(float(v["total"]) == world * B for v in r)  # whole-job instance count


def test_eight_rank_shards_of_the_segway_config(tmp_path):
    """BASELINE configs[3] as it is split over a node: 8 ranks x 32 768 agents = the 262 144-agent seeded stream of C4,
    each rank its own contiguous block, no rank's block overlapping another's (gloo on the CPU: the 8-GPU run itself is
    the driver's)."""
    B, world = 32768, 8
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path), 4), nprocs=world, join=True)
    from asif_amd import workloads
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    assert [int(v["first"]) for v in r] == [k * B for k in range(world)]
    whole_x, whole_u = workloads.make_batch(4, world * B)
    assert whole_x.shape == (4, 262144)
    assert np.array_equal(np.concatenate([v["x"] for v in r], axis=1), whole_x)
    assert np.array_equal(np.concatenate([v["u"] for v in r], axis=1), whole_u)
    assert all(float(v["tmax"]) == 8.0 for v in r) and all(float(v["total"]) == world * B for v in r)
