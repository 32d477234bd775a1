"""Worker of tests/test_gpu_model.py::test_two_gpu_rccl_exchange_modes_agree (one process per GPU, RCCL communicator over the host group's
broadcast): a few one-call training steps in exchange mode 0 (all-reduce + full update) or 1 (reduce-scatter, Adam on this rank's shard,
all-gather), every parameter printed by rank 0 for the parent to compare; the guards around rank-local moments are exercised on the way."""
import sys

import numpy as np

from deepcgp_amd import device as dev, synthetic as syn
from deepcgp_amd.dist import HostGroup, env_rank_world, init_rccl, shard_range
from deepcgp_amd.models import build_from_spec

mode = int(sys.argv[1])
rank, world, _ = env_rank_world()
grp = HostGroup(rank, world)
ctx = dev.get_context()                       # device = LOCAL_RANK
init_rccl(ctx, rank, world, grp.broadcast_bytes)
assert ctx.comm_count() == world
hwc, N = (12, 12, 1), 8
spec = syn.make_spec(hwc, [(3, 2, 4)], (3, 1), M=24, S=3, num_data=500, seed=5, conv_q_sqrt_scale=0.3)
X, Y = syn.make_batch(hwc, N, seed=5)
lo, hi = shard_range(N, rank, world)
model = build_from_spec(spec, X[lo:hi], Y[lo:hi])
model.global_batch = N
model.set_shard(lo, N)
model.set_grad_exchange(mode)
elbos = [model.train_step(X[lo:hi], Y[lo:hi], 0.01, seed=3 + i) for i in range(4)]
if mode == 1:      # moments are rank-local now: the all-reduce route and a full update must be refused, not silently diverge
    for bad in (lambda: model.set_grad_exchange(0), lambda: (model.compute_gradients(X[lo:hi], Y[lo:hi], seed=9), model.adam_step(0.01))):
        try:
            bad()
        except dev.DcgpError as e:
            assert "shard" in str(e), e
        else:
            raise AssertionError("a full update on top of rank-local moments was accepted")
model.pull_parameters()
vals = []
for li, l in enumerate(model.layers):
    head = li == len(model.layers) - 1
    kern = (l.kern.base_kernel if hasattr(l.kern, "base_kernel") else l.kern) if head else l.base_kernel
    vals += [np.ravel(l.feature.Z), np.ravel(l.q_mu), np.ravel(l.q_sqrt), np.ravel(kern.variance), np.ravel(kern.lengthscales)]
    if head and hasattr(l.kern, "patch_weights"):
        vals.append(np.ravel(l.kern.patch_weights))
flat = np.concatenate([np.asarray(v, np.float64) for v in vals])
every = grp.all_gather(flat)                   # the replicas must hold the same parameters
assert np.array_equal(every[:flat.size], every[flat.size:2 * flat.size]), "replicas differ"
grp.barrier()
if rank == 0:
    print("PARAMS " + " ".join("%.17g" % v for v in flat[::7]))
    print("ELBOS " + " ".join("%.17g" % e for e in elbos))
model.close()
grp.close()
