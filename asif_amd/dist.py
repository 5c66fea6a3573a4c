"""Multi-process plumbing of bench.py: the batch axis shards embarrassingly, so the only distributed
pieces are the rendezvous, the barrier around the timed region and the max-over-ranks of the elapsed
time.  There is no data-path collective and, with the default backend "gloo" (TCP on the host), no RCCL at all --
north_star: "no RCCL collectives".  (In-process multi-GPU: asif_hip_filter_batch_host_multi in the library.)
"""
import os

import torch


class Group:
    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = backend or "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            # gloo announces its connections on stdout; bench.py's stdout carries exactly one JSON line
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            self.dist = dist
            self.backend = backend

    def shard(self, per_rank_batch):
        """(first instance, count) of this rank's slice of the seeded instance stream: rank r owns
        [r*B, (r+1)*B) -- weak scaling, every rank the same amount of work."""
        return self.rank * per_rank_batch, per_rank_batch

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(self, value, device=None):
        if self.dist is None:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value, device=None):
        if self.dist is None:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
