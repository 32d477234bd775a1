"""Multi-GPU sharding of the forward ELBO (one process per GPU).

The data term of the ELBO is a sum over minibatch images and every layer's conditional is independent
per image, so the N images of a minibatch (each with all S samples and all P patches) are partitioned
contiguously over the ranks.  Parameters, Kuu, the Choleskys and the KL terms are replicated.  The only
exchange per step is a sum all-reduce of ONE float64 (the per-rank data term): on GPUs it is an RCCL
``ncclAllReduce`` issued on the ctx stream inside ``dcgp_elbo_forward``; the same assembly logic runs
over ``torch.distributed`` (gloo) in the CPU tests.
"""
import os

import numpy as np


def shard_range(n, rank, world):
    """Contiguous partition of range(n); the first n % world ranks get one extra element."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(X, Y, zs, rank, world):
    """This rank's images, labels and noise (z is indexed [S, image, D], so a shard sees exactly the
    rows it would see in the full batch -> results are independent of the number of ranks)."""
    lo, hi = shard_range(np.shape(X)[0], rank, world)
    zs_l = None if zs is None else [None if z is None else np.ascontiguousarray(z[:, lo:hi]) for z in zs]
    return X[lo:hi], Y[lo:hi], zs_l


def assemble_elbo(global_data_term, kl, num_data, global_batch):
    """ELBO = sum_n E_q log p(y_n) * num_data / batch - sum_l KL_l."""
    return global_data_term * (float(num_data) / float(global_batch)) - kl


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def allreduce_sum_host(value, group=None):
    """Sum a Python float over the ranks of an initialised torch.distributed group (gloo on CPU)."""
    import torch
    import torch.distributed as td
    t = torch.tensor([float(value)], dtype=torch.float64)
    td.all_reduce(t, op=td.ReduceOp.SUM, group=group)
    return float(t[0])


def init_rccl(ctx, rank, world, broadcast_bytes):
    """Create the RCCL communicator of ``ctx``: rank 0 draws the unique id, ``broadcast_bytes(b, src=0)``
    (any host-side broadcast, e.g. torch.distributed gloo) ships the 128 bytes to the other ranks."""
    from . import device as dev
    uid = dev.comm_unique_id() if rank == 0 else bytes(128)
    uid = broadcast_bytes(uid)
    ctx.comm_init(world, rank, uid)
