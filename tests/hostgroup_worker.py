"""Worker of tests/test_host_cpu.py::test_two_rank_hostgroup_elbo (one process per rank, deepcgp_amd.dist.HostGroup over TCP)."""
import numpy as np

from deepcgp_amd import synthetic as syn
from deepcgp_amd.dist import HostGroup, shard_batch, assemble_elbo, env_rank_world
from oracle_build import oracle_model

rank, world, _ = env_rank_world()
grp = HostGroup(rank, world)
uid = grp.broadcast_bytes(bytes(range(128)) if rank == 0 else b"")     # what init_rccl ships
assert uid == bytes(range(128))
hwc = (10, 10, 1)
spec = syn.make_spec(hwc, [(3, 2, 3)], (3, 1), M=6, S=2, num_data=777, seed=8, conv_q_sqrt_scale=0.3)
X, Y = syn.make_batch(hwc, 7, seed=8)
zs = syn.make_noise(spec, 7, seed=8)
model = oracle_model(spec, X, Y)            # the oracle stands in for the per-rank GPU data term on CPU
Xs, Ys, zl = shard_batch(X, Y, zs, rank, world)
local = model.data_term(Xs, Ys, zs=zl)
total = float(grp.allreduce([local], "sum")[0])
assert grp.allreduce([rank, -rank], "max").tolist() == [world - 1, 0.0]
elbo = assemble_elbo(total, model.KL(), spec["num_data"], X.shape[0])
full = model.compute_log_likelihood(X, Y, zs=zs)
assert abs(elbo - full) <= 1e-12 * abs(full), (elbo, full)
grp.barrier()
grp.close()
print("OK rank %d elbo %.17g" % (rank, elbo))
