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
# the training step's exchange mode 1 (dcgp_model_set_grad_exchange): reduce-scatter of the rank-local gradient blocks, Adam on this rank's
# shard, all-gather of the parameters == all-reduce + the full update, bit for bit; block lengths that divide, that do not, and n < world
from deepcgp_amd.dist import grad_shard_range, adam_update, sharded_adam_step
for n in (12, 1000, 1001, 1, 2, 37):
    rng0 = np.random.default_rng(n)
    p0, m0, v0 = rng0.standard_normal(n), 0.1 * rng0.standard_normal(n), rng0.random(n)
    g_ranks = [np.random.default_rng(1000 * n + r).standard_normal(n) for r in range(world)]     # what each rank's reverse pass left
    lo, hi, sh = grad_shard_range(n, world, rank)
    assert sh * world >= n and 0 <= lo <= hi <= n and (hi - lo == sh or hi == n)
    covered = sorted(grad_shard_range(n, world, r)[:2] for r in range(world))
    assert covered[0][0] == 0 and covered[-1][1] == n and all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))
    got = grp.reduce_scatter_sum(g_ranks[rank])
    g_sum = np.sum(np.stack(g_ranks), axis=0)        # np.sum over the stacked ranks: the order the group's rank 0 adds them in
    want = np.zeros(sh * world); want[:n] = g_sum
    np.testing.assert_array_equal(got, want[rank * sh:(rank + 1) * sh])
    m, v = m0.copy(), v0.copy()
    p_new = sharded_adam_step(grp, p0, g_ranks[rank], m, v, 0.01)
    pf, mf, vf = p0.copy(), m0.copy(), v0.copy()
    adam_update(pf, g_sum, mf, vf, 0.01)
    np.testing.assert_array_equal(p_new, pf)                       # every rank holds the fully updated block
    np.testing.assert_array_equal(m[lo:hi], mf[lo:hi])             # ... and the moments of its own shard
    np.testing.assert_array_equal(grp.all_gather(np.full(3, float(rank))), np.repeat(np.arange(world, dtype=float), 3))
grp.barrier()
grp.close()
print("OK rank %d elbo %.17g" % (rank, elbo))
