"""Multi-GPU layout of the hot path: independent video streams sharded over ranks, no data-path
collective; the single exchange step is a SUM all-reduce of the int64 count vector
(pos, neg, int, del per label -- deepdish.py:1141-1145 upstream) over RCCL ("nccl" backend on ROCm)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_streams(n_streams, rank, world):
    """stream s -> rank s mod world."""
    return [s for s in range(n_streams) if s % world == rank]


def reduce_counts(counts, device=None):
    """counts: int64 array [n_labels, 4] of this rank -> the sum over all ranks (every rank gets it)."""
    t = torch.tensor(np.asarray(counts, dtype=np.int64))          # a copy: the caller keeps its local counts
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
