"""GPU parity of the model-level path (one dcgp_elbo_forward call per minibatch) against the oracle, the
committed golden vectors, and -- at BASELINE.json's full sizes -- through size-independent properties."""
import glob
import os

import numpy as np
import pytest

from deepcgp_amd import synthetic as syn
from deepcgp_amd.models import build_from_spec
from deepcgp_amd.dist import shard_range, shard_batch, assemble_elbo
from oracle_build import oracle_model
from golden.make_golden import unflatten_spec

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g[0-9]*.npz")))   # (ops_*.npz: tests/test_golden_ops.py)
RTOL = 1e-9          # fp64 end to end; the north star's bar is 1e-4


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_vectors(ctx, path):
    d = np.load(path)
    spec = unflatten_spec(d)
    nl = len(spec["convs"]) + 1
    zs = [d["z%d" % i] for i in range(nl)]
    model = build_from_spec(spec, d["X"], d["Y"])
    Fs, Fm, Fv = model.propagate(d["X"], S=spec["S"], zs=zs)
    for i in range(nl):
        assert rel(Fm[i], d["Fmean%d" % i]) < RTOL, ("Fmean", i)
        assert rel(Fv[i], d["Fvar%d" % i]) < RTOL, ("Fvar", i)
        assert rel(Fs[i], d["Fs%d" % i]) < RTOL, ("Fs", i)
    elbo, data, kl = model.compute_log_likelihood(d["X"], d["Y"], zs=zs, return_parts=True)
    assert abs(elbo - float(d["elbo"])) <= RTOL * abs(float(d["elbo"]))
    assert abs(data - float(d["data_term"])) <= RTOL * abs(float(d["data_term"]))
    assert abs(kl - float(d["kl"])) <= RTOL * abs(float(d["kl"]))
    # layer-0 de-duplication is exact: the S copies of the batch are identical
    model.dedup_layer0 = True
    e_d = model.compute_log_likelihood(d["X"], d["Y"], zs=zs)
    assert e_d == elbo if spec["convs"] else abs(e_d - elbo) <= 1e-13 * abs(elbo)
    # per-layer KL through the operator API agrees with the fused path
    assert abs(model.KL() - kl) <= 1e-10 * abs(kl)
    # the training step against the committed gradient fixtures (tiled and de-duplicated first layer)
    for dedup in (False, True):
        model.dedup_layer0 = dedup
        e_g, grads = model.compute_gradients(d["X"], d["Y"], zs=zs)
        assert abs(e_g - float(d["elbo"])) <= RTOL * abs(float(d["elbo"]))
        for li, g in enumerate(grads):
            for name, val in g.items():
                want = d["grad%d_%s" % (li, name)]
                err = np.abs(val - want).max()
                assert err < 1e-7 * np.abs(want).max() or err < 1e-8, (dedup, li, name, err)
    model.close()


@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("case", ["cfg1_small", "ch_M40", "three_layer_cifar"])
def test_elbo_vs_oracle_midsize(ctx, case, white):
    if case == "cfg1_small":        # BASELINE configs[0] geometry (28x28x1, head f5 s1, P = 576, M = 32), S = 2, 4 images
        hwc, convs, head, M, N, S = (28, 28, 1), [], (5, 1), 32, 4, 2
    elif case == "ch_M40":          # configs[1] geometry with M not a multiple of 16
        hwc, convs, head, M, N, S = (28, 28, 1), [(5, 2, 10)], (5, 1), 40, 3, 2
    else:                           # configs[3] geometry (CIFAR 3-layer), small M
        hwc, convs, head, M, N, S = (32, 32, 3), [(4, 2, 10), (5, 1, 10)], (5, 1), 24, 2, 2
    spec = syn.make_spec(hwc, convs, head, M, S=S, num_data=50000, seed=42, white=white, conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(hwc, N, seed=42)
    zs = syn.make_noise(spec, N, seed=42)
    ref = oracle_model(spec, X, Y)
    model = build_from_spec(spec, X, Y)
    e, dt, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert abs(dt - ref.data_term(X, Y, zs=zs)) <= RTOL * abs(dt)
    assert abs(kl - ref.KL()) <= RTOL * max(abs(kl), 1.0)
    assert abs(e - ref.compute_log_likelihood(X, Y, zs=zs)) <= RTOL * abs(e)
    pm, pv = model.predict_y(X, S, zs=zs)
    om, ov = ref.predict_y(X, S, zs=zs)
    assert rel(pm, om) < RTOL and rel(pv, ov) < RTOL
    pbar = model.predict_proba(X, S, zs=zs)
    assert pbar.shape == (N, 10) and rel(pbar, om.mean(axis=0)) < RTOL
    model.close()


def test_predict_y_epsilon_and_accuracy_logger(ctx):
    """predict_y is one device call (forward + RobustMax quadrature); a non-default RobustMax epsilon reaches
    both it and the ELBO; AccuracyLogger (conv_gp/utils/log.py:50-67) over ragged batches equals the
    arg-max of the oracle's sample-mean probabilities."""
    from deepcgp_amd.models import AccuracyLogger
    hwc, N, S = (28, 28, 1), 7, 5
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), 32, S=S, num_data=1000, seed=5, conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(hwc, N, seed=5)
    ref = oracle_model(spec, X, Y)
    model = build_from_spec(spec, X, Y)
    for eps in (1e-3, 0.05):
        ref.likelihood.epsilon = eps
        ref.likelihood.eps_k1 = eps / (ref.likelihood.num_classes - 1.0)
        model.likelihood.epsilon = eps
        model.sync_parameters()
        zs = syn.make_noise(spec, N, seed=6)
        pm, pv = model.predict_y(X, S, zs=zs)
        om, ov = ref.predict_y(X, S, zs=zs)
        assert rel(pm, om) < RTOL and rel(pv, ov) < RTOL
        e = model.compute_log_likelihood(X, Y, zs=zs)
        assert abs(e - ref.compute_log_likelihood(X, Y, zs=zs)) <= RTOL * abs(e)
    # device RNG path: the logger's answer must equal the arg-max of its own probabilities, batch by batch
    logger = AccuracyLogger(X, Y, batch_size=3, num_samples=S)
    acc = logger(model, seed=11)
    want = 0
    for i, lo in enumerate(range(0, N, 3)):
        p, _ = model.predict_y(X[lo:lo + 3], S, seed=11 + i)
        want += int((p.mean(axis=0).argmax(axis=1) == Y.reshape(-1)[lo:lo + 3]).sum())
    assert acc == want / N
    assert model.predict_y(X[:0], S)[0].shape == (S, 0, 10)
    model.close()


def test_init_state_models_ill_conditioned(ctx):
    """Reference initial state (q_mu = 0, q_sqrt = 1e-5 chol(Kuu), conv_gp/models.py:136-138) on smooth
    images: var ~ 1e-3 from catastrophic cancellation in Kdiag - sum A^2 -- the case all-fp32 fails."""
    hwc = (28, 28, 1)
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), M=96, S=2, num_data=60000, seed=7, q_mu_random=False)
    X, Y = syn.make_batch(hwc, 3, seed=7)
    zs = syn.make_noise(spec, 3, seed=7)
    ref, model = oracle_model(spec, X, Y), build_from_spec(spec, X, Y)
    _, Fm, Fv = model.propagate(X, S=2, zs=zs)
    _, om, ov = ref.propagate(X, S=2, zs=zs)
    assert np.min(ov[0]) < 0.1            # the cancellation regime is actually reached
    for i in range(2):
        assert rel(Fm[i], om[i]) < 1e-8 and np.max(np.abs(Fv[i] - ov[i]) / np.maximum(np.abs(ov[i]), 1e-6)) < 1e-6
    e = model.compute_log_likelihood(X, Y, zs=zs)
    assert abs(e - ref.compute_log_likelihood(X, Y, zs=zs)) <= 1e-8 * abs(e)
    model.close()


def test_parameter_push_and_frozen_prior(ctx):
    hwc = (12, 12, 1)
    spec = syn.make_spec(hwc, [(3, 2, 4)], (3, 1), M=10, S=2, num_data=500, seed=5, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, 4, seed=5)
    zs = syn.make_noise(spec, 4, seed=5)
    model = build_from_spec(spec, X, Y)
    e0 = model.compute_log_likelihood(X, Y, zs=zs)
    rng = np.random.default_rng(0)
    # train-like update: Z, q_mu, hyper-parameters move; the KL prior keeps the initial Z (conv_gp/layers.py:149-152)
    spec["convs"][0]["Z"] = spec["convs"][0]["Z"] + 0.1 * rng.standard_normal(spec["convs"][0]["Z"].shape)
    spec["convs"][0]["q_mu"] = rng.standard_normal(spec["convs"][0]["q_mu"].shape)
    spec["convs"][0]["variance"], spec["convs"][0]["ls"] = 3.0, 4.0
    spec["head"]["w"] = 0.5 + rng.random(spec["head"]["w"].shape)
    l0, h = model.layers[0], model.layers[1]
    l0.feature.Z, l0.q_mu = spec["convs"][0]["Z"], spec["convs"][0]["q_mu"]
    l0.base_kernel.variance, l0.base_kernel.lengthscales = 3.0, 4.0
    h.kern.patch_weights = spec["head"]["w"]
    model.sync_parameters()
    e1 = model.compute_log_likelihood(X, Y, zs=zs)
    ref = oracle_model(spec, X, Y).compute_log_likelihood(X, Y, zs=zs)
    assert e1 != e0 and abs(e1 - ref) <= RTOL * abs(ref)
    names = [p.pathname for p in model.parameters]
    assert "DGP/layers/0/conv_kernel/base_kernel/variance" in names and "DGP/layers/1/kern/patch_weights" in names
    assert "DGP/layers/0/feature/Z" in names and "DGP/layers/1/q_sqrt" in names
    model.close()


def test_device_rng_path(ctx):
    hwc = (12, 12, 1)
    spec = syn.make_spec(hwc, [(3, 2, 4)], (3, 1), M=10, S=50, num_data=500, seed=5, conv_q_sqrt_scale=1.0)
    X, Y = syn.make_batch(hwc, 4, seed=5)
    model = build_from_spec(spec, X, Y)
    a = model.compute_log_likelihood(X, Y, seed=1)
    assert a == model.compute_log_likelihood(X, Y, seed=1) and a != model.compute_log_likelihood(X, Y, seed=2)
    Fs, Fm, Fv = model.propagate(X, S=400, seed=3)
    zhat = (Fs[0] - Fm[0]) / np.sqrt(Fv[0] + 1e-3)            # recovered standard normals
    assert abs(zhat.mean()) < 0.01 and abs(zhat.std() - 1.0) < 0.01
    assert abs(np.mean(zhat ** 3)) < 0.03 and abs(np.mean(zhat ** 4) - 3.0) < 0.1
    model.close()


def test_kl_pieces_in_the_tail_launch_match_the_kl_launches(ctx):
    """Unwhitened layers with M <= 256 take their KL pieces from the strip sums prep_solve leaves behind, added up by extra
    workgroups of the tail launch (no KL launches, no side stream); the ctx option kl_side keeps the GEMM + kl_small route on the
    side stream.  Same KL to rounding, same data term to the bit, for a conv layer with a frozen-Z prior, and against the oracle."""
    import os
    hwc = (14, 14, 1)
    spec = syn.make_spec(hwc, [(3, 2, 5)], (3, 1), M=24, S=4, num_data=700, seed=33, conv_q_sqrt_scale=0.7)
    X, Y = syn.make_batch(hwc, 6, seed=33)
    zs = syn.make_noise(spec, 6, seed=34)
    model = build_from_spec(spec, X, Y)
    spec["convs"][0]["Z"] = np.asarray(spec["convs"][0]["Z"]) * 1.1 + 0.05   # live Z != frozen Z0: the prior factor is not L
    model.layers[0].feature.Z = spec["convs"][0]["Z"]
    model.sync_parameters()
    e, data, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    ref = oracle_model(spec, X, Y).compute_log_likelihood(X, Y, zs=zs)
    assert abs(e - ref) <= RTOL * abs(ref)
    with ctx.options(kl_side=1):
        e2, data2, kl2 = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert data2 == data and abs(kl2 - kl) <= 1e-12 * abs(kl) and abs(e2 - e) <= 1e-12 * abs(e)
    assert abs(model.KL() - kl) <= 1e-10 * abs(kl)                   # the operator API, layer by layer
    tickets = [model.enqueue_log_likelihood(X, Y, zs=zs) for _ in range(3)]
    assert [model.collect_log_likelihood(t) for t in tickets] == [e] * 3
    model.close()


def test_kl_pieces_riding_the_head_conditional_are_bit_identical(ctx):
    """From ~1000 diagonal entries per layer on (M R >= 1024) the KL pieces are workgroups of the head's one-launch conditional instead of the tail
    launch (head_cond.hip; ctx option kl_no_ride keeps them in the tail): same sums in the same order -- ELBO, data term and KL to the bit, synchronous
    and in flight, head-first and conv + head, and against the oracle."""
    hwc = (12, 12, 1)
    for convs in ([], [(3, 2, 4)]):
        spec = syn.make_spec(hwc, convs, (3, 1), M=128, S=3, num_data=600, seed=41, conv_q_sqrt_scale=0.4)
        X, Y = syn.make_batch(hwc, 6, seed=41)
        zs = syn.make_noise(spec, 6, seed=42)
        model = build_from_spec(spec, X, Y)
        e, data, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
        ref = oracle_model(spec, X, Y).compute_log_likelihood(X, Y, zs=zs)
        assert abs(e - ref) <= RTOL * abs(ref)
        with ctx.options(kl_no_ride=1):
            assert model.compute_log_likelihood(X, Y, zs=zs, return_parts=True) == (e, data, kl)
        tickets = [model.enqueue_log_likelihood(X, Y, zs=zs) for _ in range(3)]
        assert [model.collect_log_likelihood(t) for t in tickets] == [e] * 3
        assert model.compute_log_likelihood(X, Y, zs=zs, return_parts=True) == (e, data, kl)
        model.close()


def test_enqueued_steps_match_the_synchronous_forward(ctx):
    """dcgp_elbo_forward_enqueue / _collect: several steps in flight (different minibatches, explicit noise or the
    device RNG) hand back bit-identical values to dcgp_elbo_forward, in order; misuse is refused."""
    from deepcgp_amd.device import DcgpError
    hwc = (14, 14, 1)
    spec = syn.make_spec(hwc, [(3, 2, 5)], (3, 1), M=20, S=4, num_data=700, seed=21, conv_q_sqrt_scale=0.5)
    model = build_from_spec(spec, *syn.make_batch(hwc, 6, seed=21))
    batches = [syn.make_batch(hwc, n, seed=30 + n) for n in (6, 3, 8, 5)]
    noise = [syn.make_noise(spec, 6, seed=1), None, syn.make_noise(spec, 8, seed=2), None]
    want = [model.compute_log_likelihood(X, Y, zs=z, seed=7 + i, return_parts=True) for i, ((X, Y), z) in enumerate(zip(batches, noise))]
    for _ in range(3):       # the ring of result slots wraps around
        tickets = [model.enqueue_log_likelihood(X, Y, zs=z, seed=7 + i) for i, ((X, Y), z) in enumerate(zip(batches, noise))]
        with pytest.raises(DcgpError):   # a fifth step in flight
            model.enqueue_log_likelihood(*batches[0])
        with pytest.raises(DcgpError):   # out of order
            model.collect_log_likelihood(tickets[1])
        with pytest.raises(DcgpError):   # the synchronous call while steps are outstanding
            model.compute_log_likelihood(*batches[0])
        got = [model.collect_log_likelihood(t, return_parts=True) for t in tickets]
        assert got == want
        with pytest.raises(DcgpError):   # nothing left to collect
            model.collect_log_likelihood(tickets[-1] + 1)
    assert model.compute_log_likelihood(*batches[1], seed=8) == want[1][0]
    model.close()


def test_enqueued_step_reports_failed_factorisation(ctx):
    hwc = (12, 12, 1)
    spec = syn.make_spec(hwc, [(3, 2, 4)], (3, 1), M=10, S=2, num_data=500, seed=5)
    X, Y = syn.make_batch(hwc, 4, seed=5)
    model = build_from_spec(spec, X, Y)
    good = model.compute_log_likelihood(X, Y, seed=1)
    Z = model.layers[0].feature.Z.copy()
    model.layers[0].feature.Z = np.full_like(Z, np.nan)
    model.sync_parameters()
    t = model.enqueue_log_likelihood(X, Y, seed=1)
    from deepcgp_amd.device import NotPositiveDefinite
    with pytest.raises(NotPositiveDefinite):
        model.collect_log_likelihood(t)
    model.layers[0].feature.Z = Z
    model.sync_parameters()
    t = model.enqueue_log_likelihood(X, Y, seed=1)
    assert model.collect_log_likelihood(t) == good
    model.close()


def test_shards_sum_to_full_batch_on_gpu(ctx):
    hwc = (28, 28, 1)
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), M=32, S=3, num_data=60000, seed=9, conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(hwc, 7, seed=9)
    zs = syn.make_noise(spec, 7, seed=9)
    model = build_from_spec(spec, X, Y)
    full, data, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    for world in (2, 4):
        tot = 0.0
        for rank in range(world):
            Xs, Ys, zl = shard_batch(X, Y, zs, rank, world)
            tot += model.compute_log_likelihood(Xs, Ys, zs=zl, return_parts=True)[1]
        assert abs(assemble_elbo(tot, kl, spec["num_data"], 7) - full) <= 1e-12 * abs(full)
    # the same with the DEVICE RNG (what bench.py runs): a shard declared with set_shard draws its noise at the elements' places in the
    # un-sharded batch, so the ranks' data terms still add up to the full batch's -- for the 3-layer stack and both de-dup settings too
    for dedup in (False, True):
        model.dedup_layer0 = dedup
        full, data, kl = model.compute_log_likelihood(X, Y, seed=11, return_parts=True)
        for world in (2, 3):
            tot = 0.0
            for rank in range(world):
                lo, hi = shard_range(7, rank, world)
                model.set_shard(lo, 7)
                tot += model.compute_log_likelihood(X[lo:hi], Y[lo:hi], seed=11, return_parts=True, scale=spec["num_data"] / 7.0)[1]
            model.set_shard(0, 0)
            assert abs(assemble_elbo(tot, kl, spec["num_data"], 7) - full) <= 1e-12 * abs(full), (dedup, world)
    model.close()


@pytest.mark.parametrize("name", ["cfg1_mnist_H_M32", "cfg2_mnist_CH_M256", "cfg2_mnist_H_M256"])
def test_full_size_properties(ctx, name):
    """BASELINE configs[0] and configs[1] at full size (M = 32 / 256, batch 32, S = 10): properties that need no oracle run."""
    spec, X, Y = syn.make_config(name)
    zs = syn.make_noise(spec, X.shape[0], seed=1)
    model = build_from_spec(spec, X, Y)
    e, data, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert np.isfinite([e, data, kl]).all() and kl > 0 and data < 0
    assert abs(e - (data * spec["num_data"] / X.shape[0] - kl)) <= 1e-12 * abs(e)
    model.dedup_layer0 = True
    e_d = model.compute_log_likelihood(X, Y, zs=zs)
    if spec["convs"]:
        assert e_d == e                   # exact de-duplication: the S copies give bit-identical rows
    else:
        assert abs(e_d - e) <= 1e-13 * abs(e)   # head-only: N rows are summed instead of S*N (other rounding order)
    model.dedup_layer0 = False
    perm = np.random.default_rng(0).permutation(X.shape[0])                     # image order is irrelevant
    e_p = model.compute_log_likelihood(X[perm], Y[perm], zs=[z[:, perm] for z in zs])
    assert abs(e_p - e) <= 1e-11 * abs(e)
    lo = model.compute_log_likelihood(X[:16], Y[:16], zs=[z[:, :16] for z in zs], return_parts=True)[1]
    hi = model.compute_log_likelihood(X[16:], Y[16:], zs=[z[:, 16:] for z in zs], return_parts=True)[1]
    assert abs((lo + hi) - data) <= 1e-11 * abs(data)                           # shard additivity
    model.close()


def test_full_size_cfg2_vs_oracle(ctx):
    """The headline configuration against the oracle itself on a reduced batch (S = 2, 4 images, full M = 256)."""
    spec, X, Y = syn.make_config("cfg2_mnist_CH_M256", S=2)
    X, Y = X[:4], Y[:4]
    zs = syn.make_noise(spec, 4, seed=2)
    model, ref = build_from_spec(spec, X, Y), oracle_model(spec, X, Y)
    e, data, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert abs(data - ref.data_term(X, Y, zs=zs)) <= 1e-8 * abs(data)
    assert abs(kl - ref.KL()) <= 1e-8 * abs(kl)
    assert abs(e - ref.compute_log_likelihood(X, Y, zs=zs)) <= 1e-8 * abs(e)
    model.close()


@pytest.mark.parametrize("hwc,conv,M,N,S", [((13, 13, 1), (4, 1, 3), 24, 5, 3),      # P = 100: ragged last fragment, replica-outer stores
                                            ((12, 12, 2), (5, 1, 2), 37, 3, 4),      # P = 64 (aligned: direct stores), M not a multiple of 16
                                            ((28, 28, 1), (5, 2, 10), 256, 4, 10),   # the headline first layer: P = 144, two column ranges
                                            ((32, 32, 3), (4, 2, 10), 384, 2, 5)])   # the CIFAR first layer: P = 225, L = 48, M > 256
def test_kuf_sweep_replica_stores_are_exact(ctx, hwc, conv, M, N, S):
    """The storing sweep evaluates rows that show the same image once (DGP_Base.propagate tiles the batch S times) and stores the tile to
    every such row -- direct, or replica-outer from held batches where P is not a multiple of 16 (csrc/head_units.hip).  The ELBO and every
    layer output on the sweep + GEMM route must be BIT-identical with that switched off (ctx option kuf_no_rep), and agree with the oracle."""
    spec = syn.make_spec(hwc, [conv], (3, 1), M=M, S=S, num_data=900, seed=61, conv_q_sqrt_scale=0.4)
    X, Y = syn.make_batch(hwc, N, seed=61)
    zs = syn.make_noise(spec, N, seed=61)
    model = build_from_spec(spec, X, Y)
    with ctx.options(no_fused_layer=1):
        e_rep = model.compute_log_likelihood(X, Y, zs=zs)
        f_rep = model.propagate(X, S=S, zs=zs)[0][0].copy()
        with ctx.options(kuf_no_rep=1):
            e_all = model.compute_log_likelihood(X, Y, zs=zs)
            f_all = model.propagate(X, S=S, zs=zs)[0][0].copy()
        for split in (0, 3):                      # other cuts of a row fragment's column fragments
            with ctx.options(kuf_split=split):
                assert model.compute_log_likelihood(X, Y, zs=zs) == e_rep
    assert e_rep == e_all
    np.testing.assert_array_equal(f_rep, f_all)
    if M <= 64:
        ref = oracle_model(spec, X, Y).compute_log_likelihood(X, Y, zs=zs)
        assert abs(e_rep - ref) <= RTOL * abs(ref)
    model.close()


def test_chain_graph_replay_matches_the_launch_loop(ctx):
    """ctx option chain_graph: the factorisation chain's panel launches (conv_gp/conditionals.py:29, layers.py:151,156) replayed from a HIP
    graph captured on first use (csrc/chol_fused.hip; off by default: measured slower).  Same kernels in the same order: the ELBO over several
    steps -- both banks, the first one capturing, the later ones replaying -- and the gradients are bit-identical to the loop's, for two
    models alive at once (two argument sets in the cache) and an M whose last panel is ragged."""
    out = {}
    for g in (0, 1, 2):   # 0: the launch loop, 1: the replayed graph -- both with G / alpha by their own launch (a graph does not carry the
        # right-hand sides that otherwise ride the chain, ctx option no_rhs_ride) -- 2: the loop with them riding (the default route)
        with ctx.options(chain_graph=1 if g == 1 else 0, no_rhs_ride=0 if g == 2 else 1):
            vals = []
            models = []
            for hwc, convs, head, M in (((13, 13, 2), [(4, 3, 7)], (2, 1), 41), ((12, 12, 1), [], (3, 1), 96)):
                spec = syn.make_spec(hwc, convs, head, M, S=2, num_data=555, seed=71, conv_q_sqrt_scale=0.3)
                X, Y = syn.make_batch(hwc, 3, seed=71)
                zs = syn.make_noise(spec, 3, seed=71)
                model = build_from_spec(spec, X, Y)
                models.append(model)
                for rep in range(4):
                    vals.append(model.compute_log_likelihood(X, Y, zs=zs))
                e, grads = model.compute_gradients(X, Y, zs=zs)
                vals.append(e)
                vals.append(float(sum(np.sum(np.abs(v)) for gl in grads for v in gl.values())))
            for m in models:
                m.close()
            out[g] = vals
    assert np.all(np.isfinite(out[1])) and out[0] == out[1], (out[0], out[1])
    # the riding right-hand sides sum the same products in another order: equal to rounding
    np.testing.assert_allclose(out[2], out[0], rtol=1e-11, atol=0)


def test_full_size_cfg1_vs_oracle(ctx):
    """BASELINE configs[0] (the reference's own CPU-runnable case: head only, M = 32, N = 1000, batch 32, S = 10) at its FULL size
    against the oracle: small enough for the NumPy restatement to finish in seconds (320 x 576 x 576 kernel values for Kdiag)."""
    spec, X, Y = syn.make_config("cfg1_mnist_H_M32")
    zs = syn.make_noise(spec, X.shape[0], seed=5)
    model, ref = build_from_spec(spec, X, Y), oracle_model(spec, X, Y)
    e, data, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert abs(data - ref.data_term(X, Y, zs=zs)) <= 1e-9 * abs(data)
    assert abs(kl - ref.KL()) <= 1e-9 * abs(kl)
    assert abs(e - ref.compute_log_likelihood(X, Y, zs=zs)) <= 1e-9 * abs(e)
    model.close()


def test_propagate_and_predict_y_vs_torch_forward(ctx):
    """DGP_Base.propagate (the head's marginals of every sample) and predict_y (RobustMax class probabilities, conv_gp/utils/log.py:62-66)
    against the torch forward: 28 x 28 geometry, conv layer + head, M = 48."""
    torch = pytest.importorskip("torch")
    from test_oracle_autograd import _torch_elbo, _robustmax_predict
    hwc, N, S = (28, 28, 1), 5, 3
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), 48, S=S, num_data=60000, seed=23, conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(hwc, N, seed=23)
    zs = syn.make_noise(spec, N, seed=23)
    model = build_from_spec(spec, X, Y)
    with torch.no_grad():
        _, _, mean_t, var_t = _torch_elbo(spec, X, Y, zs, want_head=True)
        p_t = _robustmax_predict(mean_t.reshape(S * N, -1), var_t.reshape(S * N, -1)).reshape(S, N, -1).numpy()
    Fs, Fmeans, Fvars = model.propagate(X, S=S, zs=zs)
    assert np.abs(Fmeans[-1] - mean_t.numpy()).max() <= 1e-9 * max(1.0, np.abs(mean_t.numpy()).max())
    assert np.abs(Fvars[-1] - var_t.numpy()).max() <= 1e-9 * max(1.0, np.abs(var_t.numpy()).max())
    ps, pv = model.predict_y(X, S, zs=zs)
    assert np.abs(ps - p_t).max() <= 1e-10
    assert np.abs(ps.sum(-1) - 1.0).max() < 2e-3          # RobustMax probabilities of the K classes sum to ~1 (quadrature + epsilon)
    model.close()


@pytest.mark.parametrize("white", [False, True])
def test_arccosine_model_vs_torch_forward(ctx, white):
    """ArcCosine(order 0) conv layers (--base-kernel acos, conv_gp/models.py:118-119) on the device against the torch forward, MNIST geometry."""
    torch = pytest.importorskip("torch")
    from test_oracle_autograd import _torch_elbo
    hwc, N, S = (28, 28, 1), 4, 2
    spec = syn.make_spec(hwc, [(5, 2, 4)], (5, 1), 40, S=S, num_data=60000, seed=29, white=white, conv_q_sqrt_scale=0.2, base_kernel="acos")
    X, Y = syn.make_batch(hwc, N, seed=29)
    zs = syn.make_noise(spec, N, seed=29)
    model = build_from_spec(spec, X, Y)
    e = model.compute_log_likelihood(X, Y, zs=zs)
    with torch.no_grad():
        e_t, _ = _torch_elbo(spec, X, Y, zs)
    assert abs(e - e_t.item()) <= 1e-9 * abs(e), (e, e_t.item())
    model.close()


def test_full_size_cfg1_vs_torch_forward(ctx):
    """The same configuration at full size against the independently written torch forward of tests/test_oracle_autograd.py (float64, CPU):
    BASELINE configs[0] is the one configuration the reference itself runs on a CPU, and this is the check of the device path at that size
    that owes nothing to oracle/."""
    pytest.importorskip("torch")
    from test_oracle_autograd import _torch_elbo
    spec, X, Y = syn.make_config("cfg1_mnist_H_M32")
    zs = syn.make_noise(spec, X.shape[0], seed=5)
    model = build_from_spec(spec, X, Y)
    e = model.compute_log_likelihood(X, Y, zs=zs)
    e_t, _ = _torch_elbo(spec, X, Y, zs)
    assert abs(e - e_t.item()) <= 1e-9 * abs(e), (e, e_t.item())
    model.close()


@pytest.mark.parametrize("name", ["cfg2_mnist_H_M256", "cfg2_mnist_CH_M256", "cfg3_mnist_3layer_M256", "cfg4_cifar_3layer_M384"])
def test_full_size_baseline_configs_vs_torch_forward(ctx, name):
    """BASELINE configs[1] -- the configuration the metric is quoted on, in both readings of "1-layer": M = 256, batch 32, S = 10, 46080 patch
    columns through the conv layer -- and configs[2], [3] (three layers, batch 64 / CIFAR M = 384) at FULL size against the same torch forward
    (2-5 s of CPU each; configs[4] at M = 1024 would take minutes and stays with the oracle on a reduced batch)."""
    torch = pytest.importorskip("torch")
    from test_oracle_autograd import _torch_elbo
    spec, X, Y = syn.make_config(name)
    zs = syn.make_noise(spec, X.shape[0], seed=6)
    model = build_from_spec(spec, X, Y)
    e = model.compute_log_likelihood(X, Y, zs=zs)
    with torch.no_grad():
        e_t, _ = _torch_elbo(spec, X, Y, zs)
    assert abs(e - e_t.item()) <= 1e-9 * abs(e), (e, e_t.item())
    model.close()


@pytest.mark.parametrize("name", ["cfg2_mnist_H_M256", "cfg2_mnist_CH_M256"])
def test_full_size_cfg2_gradient_vs_torch_autograd(ctx, name):
    """The training step's gradient at the FULL size of the configuration the metric is quoted on against PyTorch autograd of the torch forward:
    every parameter group of every layer (relative to the group's largest entry)."""
    torch = pytest.importorskip("torch")
    from test_oracle_autograd import _torch_elbo
    spec, X, Y = syn.make_config(name)
    zs = syn.make_noise(spec, X.shape[0], seed=6)
    model = build_from_spec(spec, X, Y)
    e, grads = model.compute_gradients(X, Y, zs=zs)
    e_t, leaves = _torch_elbo(spec, X, Y, zs)
    assert abs(e - e_t.item()) <= 1e-9 * abs(e)
    flat = [(li, k, t) for li, p in enumerate(leaves) for k, t in p.items()]
    tg = torch.autograd.grad(e_t, [t for _, _, t in flat])
    for (li, gname, _), g in zip(flat, tg):
        want, got = g.numpy(), np.asarray(grads[li][gname], np.float64)
        if gname == "q_sqrt":
            want, got = np.tril(want), np.tril(got)
        err = np.abs(got - want).max()
        assert err <= 1e-7 * max(1.0, np.abs(want).max()), (name, li, gname, err, np.abs(want).max())
    model.close()


@pytest.mark.parametrize("M", [256, 200, 129])
def test_symmetric_contraction_kernel_equals_the_general_gemm(ctx, M):
    """W_r = 2 A1 diag(gv_r) A1^T of the tiled conv layer's reverse pass (K = 46 080 columns at the headline size) runs on its own kernel
    (csrc/gemm_gen.hip: syrk_kscale_kernel); ctx option no_syrk sends it through the general GEMM.  Every gradient must agree to rounding --
    the two sum the columns in different orders.  M = 200 / 129: rows the kernel's 256-row chunk pads with zeros, a ragged last block."""
    if M == 256:
        spec, X, Y = syn.make_config("cfg2_mnist_CH_M256")
    else:
        hwc = (28, 28, 1)
        spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), M, S=10, num_data=60000, seed=5 + M, conv_q_sqrt_scale=0.3)
        X, Y = syn.make_batch(hwc, 8, seed=M)      # 8 x 10 x 144 = 11 520 columns: past the kernel's threshold
    zs = syn.make_noise(spec, X.shape[0], seed=11)
    model = build_from_spec(spec, X, Y)
    e1, g1 = model.compute_gradients(X, Y, zs=zs)
    with ctx.options(no_syrk=1):
        e0, g0 = model.compute_gradients(X, Y, zs=zs)
    assert e0 == e1
    for li, (a, b) in enumerate(zip(g1, g0)):
        for k in a:
            x, y = np.asarray(a[k], np.float64), np.asarray(b[k], np.float64)
            assert np.abs(x - y).max() <= 1e-11 * max(1.0, np.abs(y).max()), (li, k, np.abs(x - y).max(), np.abs(y).max())
    model.close()


@pytest.mark.parametrize("name", ["cfg5_mnist_H_M1024", "cfg5_mnist_CH_M1024"])
def test_cfg5_reduced_batch_vs_torch_forward(ctx, name):
    """BASELINE configs[4] (M = 1024: the 32-panel factorisation chain, the GEMM route of the conditional) on a reduced batch (4 images, S = 2)
    against the torch forward -- at its full batch that forward would take minutes of CPU."""
    torch = pytest.importorskip("torch")
    from test_oracle_autograd import _torch_elbo
    spec, X, Y = syn.make_config(name, S=2)
    X, Y = X[:4], Y[:4]
    zs = syn.make_noise(spec, 4, seed=8)
    model = build_from_spec(spec, X, Y)
    e = model.compute_log_likelihood(X, Y, zs=zs)
    with torch.no_grad():
        e_t, _ = _torch_elbo(spec, X, Y, zs)
    assert abs(e - e_t.item()) <= 1e-9 * abs(e), (e, e_t.item())
    model.close()


BIG = ["cfg3_mnist_3layer_M256", "cfg4_cifar_3layer_M384", "cfg5_mnist_H_M1024", "cfg5_mnist_CH_M1024"]


@pytest.mark.parametrize("name", BIG)
def test_baseline_configs_3_to_5_vs_oracle(ctx, name):
    """BASELINE.json configs[2..4] against the oracle at FULL M (256 x 3 layers / 384 / 1024) on a reduced batch (2 images,
    S = 2): every layer's sample / mean / variance and the three ELBO parts.  M = 384 / 1024 take other code than cfg2 (the
    generic GEMMs instead of the fused head conditional, 12- / 32-panel factorisation chains, other tile configurations)."""
    spec, X, Y = syn.make_config(name, S=2)
    X, Y = X[:2], Y[:2]
    zs = syn.make_noise(spec, 2, seed=3)
    model, ref = build_from_spec(spec, X, Y), oracle_model(spec, X, Y)
    oFs, oFm, oFv = ref.propagate(X, S=2, zs=zs)
    Fs, Fm, Fv = model.propagate(X, S=2, zs=zs)
    for i in range(len(Fs)):
        assert rel(Fm[i], oFm[i]) < 1e-8, ("Fmean", i, rel(Fm[i], oFm[i]))
        assert rel(Fv[i], oFv[i]) < 1e-8, ("Fvar", i, rel(Fv[i], oFv[i]))
        assert rel(Fs[i], oFs[i]) < 1e-8, ("Fs", i)
    e, data, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert abs(data - ref.data_term(X, Y, zs=zs)) <= 1e-8 * abs(data)
    assert abs(kl - ref.KL()) <= 1e-8 * abs(kl)
    assert abs(e - ref.compute_log_likelihood(X, Y, zs=zs)) <= 1e-8 * abs(e)
    model.dedup_layer0 = True
    e_d = model.compute_log_likelihood(X, Y, zs=zs)
    assert abs(e_d - e) <= 1e-13 * abs(e)
    model.close()


@pytest.mark.parametrize("name", BIG)
def test_baseline_configs_3_to_5_full_size_properties(ctx, name):
    """The same configurations at their full batch (64 / 32 / 128 images, S = 10) through properties that need no oracle run:
    ELBO assembly, exact layer-0 de-duplication, image-order invariance, shard additivity (the multi-GPU decomposition), and
    agreement of the synchronous and the enqueued step."""
    spec, X, Y = syn.make_config(name)
    N = X.shape[0]
    zs = syn.make_noise(spec, N, seed=1)
    model = build_from_spec(spec, X, Y)
    e, data, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert np.isfinite([e, data, kl]).all() and kl > 0 and data < 0
    assert abs(e - (data * spec["num_data"] / N - kl)) <= 1e-12 * abs(e)
    model.dedup_layer0 = True
    e_d = model.compute_log_likelihood(X, Y, zs=zs)
    assert (e_d == e) if spec["convs"] else abs(e_d - e) <= 1e-13 * abs(e)
    model.dedup_layer0 = False
    perm = np.random.default_rng(0).permutation(N)
    e_p = model.compute_log_likelihood(X[perm], Y[perm], zs=[z[:, perm] for z in zs])
    assert abs(e_p - e) <= 1e-11 * abs(e)
    cut = N // 4 + 1                                                            # ragged shards
    lo = model.compute_log_likelihood(X[:cut], Y[:cut], zs=[z[:, :cut] for z in zs], return_parts=True)[1]
    hi = model.compute_log_likelihood(X[cut:], Y[cut:], zs=[z[:, cut:] for z in zs], return_parts=True)[1]
    assert abs((lo + hi) - data) <= 1e-11 * abs(data)
    t = model.enqueue_log_likelihood(X, Y, zs=zs)
    assert model.collect_log_likelihood(t) == e
    model.close()


def test_oversized_operand_slab(ctx):
    """[M x columns] = 1024 x 288 000 doubles = 2.36 GB.  The one-launch layer (conv_fused.hip) never materialises it and
    takes the size in its stride.  The sweep + GEMM route fetches whole k-tiles through 32-bit-offset buffer descriptors: a
    forward pass takes the batch in chunks of whole images (same noise per column, so the two routes agree), and the
    training step -- whose reverse pass needs K_uf and A1 of the whole batch -- must REFUSE an operand of 2 GiB or more with
    DCGP_ERR_ARG and a message naming the limit, never wrap silently."""
    import os
    from deepcgp_amd import device as dev
    spec, X, Y = syn.make_config("cfg5_mnist_CH_M1024", S=10)
    X = np.tile(X, (2, 1))[:200]                    # 200 images x 10 samples x 144 patches
    Y = np.tile(Y, 2)[:200]
    model = build_from_spec(spec, X, Y)
    with ctx.options(fused_large=1):              # M > 256 takes the one-launch route on request only
        e, data, kl = model.compute_log_likelihood(X, Y, seed=0, return_parts=True)
        lo = model.compute_log_likelihood(X[:100], Y[:100], seed=0, return_parts=True)[1]
    assert np.isfinite([e, data, kl]).all() and data < lo < 0
    e2, data2, kl2 = model.compute_log_likelihood(X, Y, seed=0, return_parts=True)      # default route: chunked
    assert abs(data2 - data) <= 1e-8 * abs(data) and abs(kl2 - kl) <= 1e-10 * abs(kl)
    lo2 = model.compute_log_likelihood(X[:100], Y[:100], seed=0, return_parts=True)[1]   # under the limit: one chunk
    assert abs(lo2 - lo) <= 1e-8 * abs(lo)
    with pytest.raises(dev.DcgpError) as ei:
        model.compute_gradients(X, Y, seed=0)
    assert ei.value.code == dev.ERR_ARG and "2 GiB" in str(ei.value)
    model.close()


def test_rccl_single_rank_allreduce_path(ctx):
    """The N>1 code path on one GPU: a 1-rank RCCL communicator, the in-stream all-reduce inside
    dcgp_elbo_forward, the per-layer gradient all-reduce inside dcgp_elbo_grad and the explicit
    dcgp_allreduce_sum_f64 entry point."""
    from deepcgp_amd import device as dev
    hwc = (12, 12, 1)
    spec = syn.make_spec(hwc, [(3, 2, 4)], (3, 1), M=10, S=2, num_data=500, seed=5, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, 4, seed=5)
    zs = syn.make_noise(spec, 4, seed=5)
    model = build_from_spec(spec, X, Y)
    e0 = model.compute_log_likelihood(X, Y, zs=zs)
    _, g0 = model.compute_gradients(X, Y, zs=zs)
    ctx.comm_init(1, 0, dev.comm_unique_id())
    try:
        assert model.compute_log_likelihood(X, Y, zs=zs) == e0
        tickets = [model.enqueue_log_likelihood(X, Y, zs=zs) for _ in range(3)]    # steps in flight, each with its all-reduce
        assert [model.collect_log_likelihood(t) for t in tickets] == [e0] * 3
        # training step: one in-stream all-reduce per layer over its contiguous gradient block (identity with one rank)
        e1, g1 = model.compute_gradients(X, Y, zs=zs)
        assert e1 == e0
        for a, b in zip(g0, g1):
            for name in a:
                np.testing.assert_array_equal(a[name], b[name])
        buf = ctx.to_device(np.array([1.5, -2.0, 3.25]))
        ctx.allreduce_sum(buf)
        np.testing.assert_array_equal(buf.numpy(), [1.5, -2.0, 3.25])
    finally:
        dev.lib().dcgp_comm_destroy(ctx.handle)
    model.close()


def test_model_builder_from_flags(ctx):
    """ModelBuilder with the reference's flag names (conv_gp/models.py:43-70): k-means patch init, identity-conv
    propagation of the init images, q_sqrt scaled by 1e-5; the built model evaluates and its initial KL is ~0 for the
    head and finite for the conv layer."""
    from deepcgp_amd.arguments import default_parser
    from deepcgp_amd.models import ModelBuilder
    rng = np.random.default_rng(0)
    np.random.seed(0)
    X = rng.standard_normal((40, 12, 12, 1))
    Y = rng.integers(0, 10, (40, 1))
    flags = default_parser().parse_args(['--name', 't', '-M', '6,7', '--feature-maps', '3', '--filter-sizes', '3,3',
                                         '--strides', '2,1', '--num-samples', '2', '--batch-size', '8'])
    model = ModelBuilder(flags, X, Y).build()
    assert [type(l).__name__ for l in model.layers] == ['ConvLayer', 'SVGP_Layer']
    conv, head = model.layers
    assert conv.num_outputs == 5 * 5 * 3 and conv.feature.Z.shape == (6, 9) and head.feature.Z.shape == (7, 27)
    assert np.max(np.abs(conv.q_sqrt)) < 1e-3 and head.q_sqrt.shape == (10, 7, 7)
    e = model.compute_log_likelihood(X[:8].reshape(8, -1), Y[:8], seed=1)
    assert np.isfinite(e)
    assert abs(head.KL()) < 1e-8 and np.isfinite(conv.KL()) and conv.KL() > 0
    # checkpoint round trip in the reference's format (experiment.py:56-64 writes it, models.py:200-240 reads it):
    # perturb the parameters, save, rebuild from the file -> same parameters, same ELBO
    import tempfile
    from deepcgp_amd.models import save_model_parameters
    conv.q_mu = rng.standard_normal(conv.q_mu.shape)
    head.q_mu = rng.standard_normal(head.q_mu.shape)
    head.kern.patch_weights = 1.0 + 0.1 * rng.standard_normal(head.kern.patch_weights.shape)
    conv.base_kernel.variance, head.kern.base_kernel.lengthscales = 3.5, 4.25
    model.sync_parameters()
    e1 = model.compute_log_likelihood(X[:8].reshape(8, -1), Y[:8], seed=1)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 't.npy')
        saved = save_model_parameters(model, path, global_step=123)
        assert 'DGP/layers/0/conv_kernel/base_kernel/variance' in saved and 'DGP/layers/1/kern/patch_weights' in saved
        np.random.seed(0)
        flags.load_model = 't'      # --load-model NAME (conv_gp/models.py:46-47); the experiment resolves the path
        b2 = ModelBuilder(flags, X, Y, model_path=path)
        m2 = b2.build()
        flags.load_model = None
    assert b2.global_step == 123
    for p1, p2 in zip(model.parameters, m2.parameters):
        assert p1.pathname == p2.pathname and np.array_equal(np.asarray(p1.value), np.asarray(p2.value)), p1.pathname
    e2 = m2.compute_log_likelihood(X[:8].reshape(8, -1), Y[:8], seed=1)
    assert e2 == e1 and e1 != e
    # the reference's loop on the rebuilt model: a few optimisation steps from the flags' batch size, accuracy logger,
    # checkpoint of the trained values (the README snippet)
    from deepcgp_amd.models import train, AccuracyLogger
    hist = train(m2, 4, lr=0.01, seed=2)
    assert len(hist) == 4 and all(np.isfinite(hist))
    acc = AccuracyLogger(X[:20], Y[:20])(m2)
    assert 0.0 <= acc <= 1.0
    with tempfile.TemporaryDirectory() as tmp:
        saved2 = save_model_parameters(m2, os.path.join(tmp, 't2.npy'), global_step=127)
    assert not np.array_equal(saved2['DGP/layers/1/q_mu'], saved['DGP/layers/1/q_mu'])      # the parameters moved
    m2.close()
    with pytest.raises(AssertionError):
        flags.feature_maps = '3,3'
        ModelBuilder(flags, X, Y).build()
    model.close()


def test_reference_format_checkpoint_loads_and_evaluates(ctx):
    """A checkpoint in the reference's own key set (tests/golden/checkpoint/ref_checkpoint_3layer.npy: the path names of
    notebooks/Inspect.ipynb cell 6 + global_step) -- not a file this repo wrote -- through ModelBuilder(--load-model): every
    parameter lands where the file says, the model evaluates to the oracle's ELBO and layer moments for those parameters, and
    writing it back gives the same key set."""
    import tempfile
    from deepcgp_amd.arguments import default_parser
    from deepcgp_amd.models import ModelBuilder, save_model_parameters
    here = os.path.join(os.path.dirname(__file__), "golden", "checkpoint")
    path = os.path.join(here, "ref_checkpoint_3layer.npy")
    raw = np.load(path, allow_pickle=True).item()
    d = np.load(os.path.join(here, "ref_checkpoint_3layer_expected.npz"))
    flags = default_parser().parse_args([str(a) for a in d["flags"]])
    flags.load_model = "fixture"
    b = ModelBuilder(flags, np.zeros((int(d["num_data"]), 14, 14, 1)), np.zeros((int(d["num_data"]), 1), np.int64), model_path=path)
    model = b.build()
    assert b.global_step == 25000 and [type(l).__name__ for l in model.layers] == ['ConvLayer', 'ConvLayer', 'SVGP_Layer']
    for p in model.parameters:
        assert p.pathname in raw, p.pathname
        np.testing.assert_array_equal(np.asarray(p.value), np.asarray(raw[p.pathname]), err_msg=p.pathname)
    zs = [d["z%d" % i] for i in range(3)]
    e, data, kl = model.compute_log_likelihood(d["X"], d["Y"], zs=zs, return_parts=True)
    assert abs(e - float(d["elbo"])) <= 1e-9 * abs(float(d["elbo"]))
    assert abs(data - float(d["data_term"])) <= 1e-9 * abs(float(d["data_term"])) and abs(kl - float(d["kl"])) <= 1e-9 * abs(float(d["kl"]))
    _, Fm, Fv = model.propagate(d["X"], S=2, zs=zs)
    for i in range(3):
        assert rel(Fm[i], d["Fmean%d" % i]) < 1e-9 and rel(Fv[i], d["Fvar%d" % i]) < 1e-9, i
    with tempfile.TemporaryDirectory() as tmp:
        saved = save_model_parameters(model, os.path.join(tmp, "w.npy"), global_step=b.global_step)
    assert set(saved) == set(raw)
    model.close()


def test_elbo_acos_base_kernel(ctx):
    """--base-kernel acos (conv_gp/models.py:118-119): ArcCosine(order=0) conv layer + RBF ConvKernel head, whole ELBO and
    layer moments against the oracle.  1e-8: acos() near cos = 1 is ill-conditioned in both implementations."""
    hwc = (28, 28, 1)
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), M=48, S=2, num_data=60000, seed=21, base_kernel="acos", conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(hwc, 4, seed=21)
    zs = syn.make_noise(spec, 4, seed=21)
    ref, model = oracle_model(spec, X, Y), build_from_spec(spec, X, Y)
    e, dt, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert abs(dt - ref.data_term(X, Y, zs=zs)) <= 1e-8 * abs(dt)
    assert abs(kl - ref.KL()) <= 1e-8 * max(abs(kl), 1.0)
    assert abs(e - ref.compute_log_likelihood(X, Y, zs=zs)) <= 1e-8 * abs(e)
    _, Fm, Fv = model.propagate(X, S=2, zs=zs)
    _, om, ov = ref.propagate(X, S=2, zs=zs)
    # layer moments: 1e-6.  Kuu's diagonal is variance * (1 - acos(1 - 1e-15) / pi): a 1-ulp difference in cos moves it
    # by ~7e-10, which inv(Kuu) (jitter 1e-3) can amplify a thousandfold -- conditioning of the reference formula
    for i in range(2):
        assert rel(Fm[i], om[i]) < 1e-6 and rel(Fv[i], ov[i]) < 1e-6
    # parameter push: change the kernel's parameters, the device copy follows
    model.layers[0].base_kernel.weight_variances = 0.6
    model.layers[0].base_kernel.bias_variance = 0.4
    ref.layers[0].conv_kernel.base_kernel.weight_variances = 0.6
    ref.layers[0].conv_kernel.base_kernel.bias_variance = 0.4
    model.sync_parameters()
    e2 = model.compute_log_likelihood(X, Y, zs=zs)
    assert abs(e2 - ref.compute_log_likelihood(X, Y, zs=zs)) <= 1e-8 * abs(e2) and e2 != e
    model.close()


@pytest.mark.parametrize("white", [False, True])
def test_elbo_dense_rbf_ard_head(ctx, white):
    """--last-kernel rbf (conv_gp/models.py:160-168): conv layer + dense RBF(ARD=True) head on the flattened 12 x 12 x 10
    features (D = 1440), one lengthscale per dimension, through the fused model path."""
    hwc = (28, 28, 1)
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), M=40, S=2, num_data=60000, seed=33, head_kernel="rbf", white=white,
                         conv_q_sqrt_scale=0.2, ls=5.0)
    spec["head"]["ls_ard"] = spec["head"]["ls_ard"] * 6.0      # sqrt(D)-ish scale so that the head kernel is not ~delta
    X, Y = syn.make_batch(hwc, 4, seed=33)
    zs = syn.make_noise(spec, 4, seed=33)
    ref, model = oracle_model(spec, X, Y), build_from_spec(spec, X, Y)
    e, dt, kl = model.compute_log_likelihood(X, Y, zs=zs, return_parts=True)
    assert abs(dt - ref.data_term(X, Y, zs=zs)) <= RTOL * abs(dt)
    assert abs(kl - ref.KL()) <= RTOL * max(abs(kl), 1.0)
    assert abs(e - ref.compute_log_likelihood(X, Y, zs=zs)) <= RTOL * abs(e)
    pm, pv = model.predict_y(X, 2, zs=zs)
    om, ov = ref.predict_y(X, 2, zs=zs)
    assert rel(pm, om) < RTOL and rel(pv, ov) < RTOL
    # the lengthscales are parameters: change them on both sides, push, compare again
    new_ls = spec["head"]["ls_ard"] * 1.3
    model.layers[-1].kern.lengthscales = new_ls.copy()
    ref.layers[-1].kern.lengthscales = new_ls.copy()
    model.sync_parameters()
    e2 = model.compute_log_likelihood(X, Y, zs=zs)
    assert abs(e2 - ref.compute_log_likelihood(X, Y, zs=zs)) <= RTOL * abs(e2) and e2 != e
    assert [p.pathname for p in model.parameters if "/layers/1/kern" in p.pathname] == ["DGP/layers/1/kern/variance", "DGP/layers/1/kern/lengthscales"]
    model.close()


@pytest.mark.parametrize("white,additive,idmean", [(False, False, False), (True, False, False), (False, True, False), (False, False, True),
                                                   (False, "dense", False), (True, "dense", False), (False, "acos", False), (True, "acos", False)])
def test_gradients_match_oracle(ctx, white, additive, idmean):
    """dcgp_elbo_grad (csrc/grad.hip) against oracle/grad.py -- itself pinned by finite differences on CPU -- on a
    three-layer model: every parameter group of every layer."""
    from oracle.grad import elbo_and_grad
    hwc, N, S = (14, 14, 1), 3, 2
    convs = [(3, 1, 3), (3, 2, 2)] if idmean else [(3, 1, 3), (4, 2, 2)]      # Conv2dMean needs odd filters
    dense = additive == "dense"               # RBF(ARD=True) head on the flattened features (--last-kernel rbf)
    acos = additive == "acos"                 # ArcCosine(order 0) base kernel on the conv layers (--base-kernel acos)
    additive = additive is True
    spec = syn.make_spec(hwc, convs, (3, 1), 20, S=S, num_data=500, seed=9, white=white,
                         conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5, head_kernel="rbf" if dense else "conv",
                         base_kernel="acos" if acos else "rbf")
    rng = np.random.default_rng(9)
    spec["head"]["w"] = 0.5 + rng.random(spec["head"]["w"].shape)
    if additive:
        spec["head"]["kernel"] = "add"
    if idmean:
        for c in spec["convs"]:
            c["mean_function"] = "conv2d"
    X, Y = syn.make_batch(hwc, N, seed=9)
    zs = syn.make_noise(spec, N, seed=9)
    ref = oracle_model(spec, X, Y)
    if additive:
        from oracle.kernels import AdditivePatchKernel
        k = ref.layers[-1].kern
        ref.layers[-1].kern = AdditivePatchKernel(k.base_kernel, k.view, k.patch_weights)
    model = build_from_spec(spec, X, Y)
    eo, go = elbo_and_grad(ref, X, Y, zs)
    import os
    # twice: the launch-per-product reverse pass of the conditional (few columns), then its one-launch strip form
    # (csrc/conv_bwd_fused.hip, taken from 4096 columns on; unwhitened layers with q_sqrt) forced onto these sizes
    for min_cols in (-1, 0):
        with ctx.options(fused_bwd_min_cols=min_cols):
            e, grads = model.compute_gradients(X, Y, zs=zs)
        assert abs(e - eo) <= RTOL * abs(eo)
        for li, (g, o) in enumerate(zip(grads, go)):
            for name, val in o.items():
                # relative to the largest entry, or 1e-8 absolute where data and KL parts cancel to a tiny net gradient
                err = np.abs(g[name] - val).max()
                assert err < 1e-7 * np.abs(val).max() or err < 1e-8, (min_cols, li, name, err, np.abs(val).max())
    model.close()


@pytest.mark.parametrize("head_kernel,white", [("conv", False), ("rbf", False), ("conv", True)])
def test_gradients_repeat_in_steady_state(ctx, head_kernel, white):
    """The reverse pass runs on three streams (csrc/grad.hip, Lanes) beside work that started during the forward pass; every
    buffer two of them touch is ordered by an event.  A missing one shows up only once nothing allocates any more: forty
    steps back to back, alternating the two forms of the conditional's reverse pass, must give the same bits as the first
    two -- and the same numbers as the single-stream order (option grad_nofork)."""
    hwc, N, S = (14, 14, 1), 3, 2
    spec = syn.make_spec(hwc, [(3, 1, 3), (4, 2, 2)], (3, 1), 20, S=S, num_data=500, seed=9, white=white, conv_q_sqrt_scale=0.3,
                         variance=2.0, ls=1.5, head_kernel=head_kernel)
    X, Y = syn.make_batch(hwc, N, seed=9)
    zs = syn.make_noise(spec, N, seed=9)
    model = build_from_spec(spec, X, Y)
    first = {}
    for it in range(40):
        mc = 0 if it % 2 else -1
        with ctx.options(fused_bwd_min_cols=mc):
            e, grads = model.compute_gradients(X, Y, zs=zs)
        if mc not in first:
            first[mc] = (e, grads)
            continue
        e0, g0 = first[mc]
        assert e == e0
        for li, (g, o) in enumerate(zip(grads, g0)):
            for name in o:
                assert np.array_equal(g[name], o[name]), (it, li, name, np.abs(g[name] - o[name]).max())
    for mc in (-1, 0):
        with ctx.options(fused_bwd_min_cols=mc, grad_nofork=1):
            e, grads = model.compute_gradients(X, Y, zs=zs)
        e0, g0 = first[mc]
        assert abs(e - e0) <= 1e-12 * abs(e0)
        for li, (g, o) in enumerate(zip(grads, g0)):
            for name in o:
                err = np.abs(g[name] - o[name]).max()
                assert err <= 1e-9 * max(1.0, np.abs(o[name]).max()), (mc, li, name, err)
    model.close()


@pytest.mark.parametrize("hwc,convs,M,N", [((9, 9, 10), [], 32, 6), ((12, 12, 10), [], 48, 5), ((14, 14, 1), [(4, 1, 10)], 32, 4),
                                           ((9, 9, 10), [], 256, 40)])
def test_kept_patch_responses_match_recomputed(ctx, hwc, convs, M, N):
    """A training step's forward sweep of a long-patch head also stores every patch response k(z_m, x_np) for the reverse pass
    (head_units_kernel<..., 3, ...>, conv_gp/kernels.py:117-133 differentiated); option grad_no_keep_k makes the reverse pass
    evaluate them again (patch_rbf).  Same ELBO bits, gradients equal to rounding -- ragged patch counts (25, 64, 49) included."""
    spec = syn.make_spec(hwc, convs, (5, 1), M, S=2, num_data=500, seed=3)
    X, Y = syn.make_batch(hwc, N, seed=3)
    model = build_from_spec(spec, X, Y)
    res = {}
    for nk in (1, 0):
        with ctx.options(grad_no_keep_k=nk):
            res[nk] = model.compute_gradients(X, Y, seed=5)
    (e1, g1), (e0, g0) = res[1], res[0]
    assert e0 == e1
    for li, (a, b) in enumerate(zip(g0, g1)):
        for k in a:
            assert np.all(np.isfinite(a[k])) and np.all(np.isfinite(b[k])), (li, k)
            assert np.abs(a[k] - b[k]).max() <= 1e-9 * max(1.0, np.abs(b[k]).max()), (li, k)
    model.close()


def test_gradients_finite_and_repeatable_at_cfg3_size(ctx):
    """BASELINE configs[2] at full size: three layers, 640 rows, a head of 25 patches of 250 elements -- far-apart patch pairs whose
    kernel values are denormals (a NaN lengthscale gradient once: 0 x log 0 in the kernel adjoint).  Finite, and the same bits twice."""
    cfg = syn.CONFIGS["cfg3_mnist_3layer_M256"]
    spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=10, num_data=cfg["num_data"], seed=1)
    X, Y = syn.make_batch(cfg["hwc"], cfg["batch"], seed=1)
    model = build_from_spec(spec, X, Y)
    model.dedup_layer0 = True
    e0, g0 = model.compute_gradients(X, Y, seed=3)
    e1, g1 = model.compute_gradients(X, Y, seed=3)
    assert np.isfinite(e0) and e0 == e1
    for li, (a, b) in enumerate(zip(g0, g1)):
        for k in a:
            assert np.all(np.isfinite(a[k])), (li, k)
            assert np.array_equal(a[k], b[k]), (li, k)
    model.close()


@pytest.mark.parametrize("white,variant", [(False, "conv"), (True, "conv"), (False, "three_layers_stride2"), (False, "additive"), (False, "dense_ard"),
                                           (True, "dense_ard"), (False, "conv2d_mean")])
def test_device_gradient_matches_torch_autograd(ctx, white, variant):
    """dcgp_elbo_grad against PyTorch autograd (CPU, float64) of the independently written textbook forward in
    tests/test_oracle_autograd.py -- third-party differentiation of a forward that shares no code with oracle/ or csrc/: conv head, additive
    head, dense RBF(ARD) head, Conv2dMean, three layers with a stride-2 first layer, both whitenings."""
    torch = pytest.importorskip("torch")
    from test_oracle_autograd import _torch_elbo
    hwc, N, S = ((14, 14, 1) if variant == "three_layers_stride2" else (10, 10, 1)), 3, 2
    convs = [(4, 2, 2), (3, 1, 2)] if variant == "three_layers_stride2" else [(3, 1, 2)]
    spec = syn.make_spec(hwc, convs, (3, 1), 7, S=S, num_data=200, seed=11, white=white, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5,
                         head_kernel="rbf" if variant == "dense_ard" else "conv")
    rng = np.random.default_rng(11)
    if variant != "dense_ard":
        spec["head"]["w"] = 0.5 + rng.random(spec["head"]["w"].shape)
    if variant == "additive":
        spec["head"]["kernel"] = "add"
    if variant == "conv2d_mean":
        spec["convs"][0]["mean_function"] = "conv2d"
    X, Y = syn.make_batch(hwc, N, seed=11)
    zs = syn.make_noise(spec, N, seed=11)
    model = build_from_spec(spec, X, Y)
    e, grads = model.compute_gradients(X, Y, zs=zs)
    e_t, leaves = _torch_elbo(spec, X, Y, zs)
    assert abs(e - e_t.item()) <= 1e-9 * abs(e)
    flat = [(li, k, t) for li, p in enumerate(leaves) for k, t in p.items()]
    tg = torch.autograd.grad(e_t, [t for _, _, t in flat])
    for (li, name, _), g in zip(flat, tg):
        want, got = g.numpy(), np.asarray(grads[li][name], np.float64)
        if name == "q_sqrt":
            want, got = np.tril(want), np.tril(got)
        err = np.abs(got - want).max()
        assert err <= 1e-7 * max(1.0, np.abs(want).max()), (variant, li, name, err)
    model.close()


def test_device_elbo_matches_torch_forward_mnist_geometry(ctx):
    """The forward ELBO at the headline geometry (28 x 28 x 1, 5 x 5 stride-2 conv layer with 10 maps, 5 x 5 conv head; M = 64, 6 images,
    S = 3) against the torch textbook forward of tests/test_oracle_autograd.py -- a forward that shares no code with oracle/ or csrc/."""
    pytest.importorskip("torch")
    from test_oracle_autograd import _torch_elbo
    hwc, N, S = (28, 28, 1), 6, 3
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), 64, S=S, num_data=60000, seed=17, conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(hwc, N, seed=17)
    zs = syn.make_noise(spec, N, seed=17)
    model = build_from_spec(spec, X, Y)
    e = model.compute_log_likelihood(X, Y, zs=zs)
    e_t, _ = _torch_elbo(spec, X, Y, zs)
    assert abs(e - e_t.item()) <= 1e-9 * abs(e), (e, e_t.item())
    model.close()


def _softplus_inv(x):
    return np.log(np.expm1(x - 1e-6))


@pytest.mark.parametrize("head_kernel,one_call", [("conv", False), ("rbf", False), ("conv+acos", False), ("conv", True), ("rbf", True)])
def test_adam_steps_match_numpy_on_oracle_gradients(ctx, head_kernel, one_call):
    """Three device Adam steps (dcgp_model_adam_step, or the one-call training step dcgp_model_train_step_adam) against
    tf.train.AdamOptimizer's update written out in numpy on the ORACLE's gradients, in gpflow's unconstrained space
    (softplus + 1e-6 for variance / lengthscales)."""
    from oracle.grad import elbo_and_grad
    from oracle_build import oracle_param_handles
    hwc, N, S, lr = (12, 12, 1), 3, 2, 0.05
    base_kernel = "acos" if head_kernel.endswith("+acos") else "rbf"      # ArcCosine(order 0) on the conv layer
    head_kernel = head_kernel.split("+")[0]
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 12, S=S, num_data=200, seed=4, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5,
                         head_kernel=head_kernel, base_kernel=base_kernel)
    X, Y = syn.make_batch(hwc, N, seed=4)
    ref = oracle_model(spec, X, Y)
    model = build_from_spec(spec, X, Y)
    handles = oracle_param_handles(ref)
    state = {(li, name): [np.zeros_like(np.array(get(), np.float64)), np.zeros_like(np.array(get(), np.float64))]
             for li, name, get, _ in handles}
    b1, b2, eps = 0.9, 0.999, 1e-8
    for t in range(1, 4):
        zs = syn.make_noise(spec, N, seed=100 + t)
        if one_call:
            e = model.train_step(X, Y, lr, zs=zs, t=t)
        else:
            e, _ = model.compute_gradients(X, Y, zs=zs, fetch=False)
            model.adam_step(lr, t)
        eo, go = elbo_and_grad(ref, X, Y, zs)
        assert abs(e - eo) <= 1e-8 * abs(eo)
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        for li, name, get, set_ in handles:
            x = np.array(get(), np.float64)
            g = -np.asarray(go[li][name], np.float64)
            positive = name in ("variance", "lengthscales", "weight_variances", "bias_variance")
            u = _softplus_inv(x) if positive else x
            if positive:
                g = g * (1.0 - np.exp(-(x - 1e-6)))
            m, v = state[(li, name)]
            m[...] = b1 * m + (1 - b1) * g
            v[...] = b2 * v + (1 - b2) * g * g
            u = u - lr_t * m / (np.sqrt(v) + eps)
            set_(np.log1p(np.exp(u)) + 1e-6 if positive else u)
    model.pull_parameters()
    for li, l in enumerate(model.layers):
        o = ref.layers[li]
        head = li == len(model.layers) - 1
        dense = head and head_kernel == "rbf"
        kern, okern = ((l.kern, o.kern) if dense else (l.kern.base_kernel, o.kern.base_kernel)) if head else (l.base_kernel, o.base_kernel)
        assert rel(l.feature.Z, o.Z) < 1e-7 and rel(l.q_mu, o.q_mu) < 1e-7 and rel(l.q_sqrt, o.q_sqrt) < 1e-7
        assert abs(kern.variance - okern.variance) < 1e-8 * okern.variance
        for pname in (("lengthscales",) if hasattr(okern, "lengthscales") else ("weight_variances", "bias_variance")):
            a, b = np.asarray(getattr(kern, pname)), np.asarray(getattr(okern, pname))
            assert np.all(np.abs(a - b) < 1e-8 * b), pname
        if head and not dense:
            assert rel(l.kern.patch_weights, o.kern.patch_weights) < 1e-7
    # and the training loop mirror runs and improves the bound on a fixed batch
    from deepcgp_amd.models import train
    model.minibatch_size = N
    hist = train(model, 20, lr=0.02, seed=1)
    assert len(hist) == 20 and np.mean(hist[-5:]) > np.mean(hist[:5])
    model.close()


def test_gradient_of_batch_shards_sums_to_full_batch(ctx):
    """Data parallelism of the training step: with the KL term weighted 1 / shards on every shard, the shard
    gradients add up to the full-batch gradient (what the per-layer all-reduce in dcgp_elbo_grad computes)."""
    hwc, N, S = (12, 12, 1), 4, 2
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 12, S=S, num_data=200, seed=6, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5)
    X, Y = syn.make_batch(hwc, N, seed=6)
    zs = syn.make_noise(spec, N, seed=6)
    model = build_from_spec(spec, X, Y)
    scale = 200.0 / N
    e_full, g_full = model.compute_gradients(X, Y, zs=zs, scale=scale, shards=1)
    parts = []
    for lo, hi in ((0, 1), (1, 4)):                      # ragged shards
        parts.append(model.compute_gradients(X[lo:hi], Y[lo:hi], zs=[z[:, lo:hi] for z in zs], scale=scale, shards=2))
    for li, g in enumerate(g_full):
        for name, val in g.items():
            tot = parts[0][1][li][name] + parts[1][1][li][name]
            err = np.abs(tot - val).max()
            assert err < 1e-9 * max(np.abs(val).max(), 1.0), (li, name, err)
    model.close()


def test_gradients_match_oracle_mnist_geometry(ctx):
    """The headline geometry (28 x 28, 5 x 5 stride-2 conv layer with 10 maps, conv head) at M = 136 -- not a
    multiple of 16 or 128 -- and 4320 patch columns: the 128 x 128 split-k products, the stacked-k launches and
    the padded rows of the gemm_tn paths are all live."""
    from oracle.grad import elbo_and_grad
    hwc, N, S = (28, 28, 1), 15, 2
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), 136, S=S, num_data=60000, seed=21, conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(hwc, N, seed=21)
    zs = syn.make_noise(spec, N, seed=21)
    ref = oracle_model(spec, X, Y)
    model = build_from_spec(spec, X, Y)
    eo, go = elbo_and_grad(ref, X, Y, zs)
    import os
    for min_cols in (-1, 0):     # second pass: the strip form of the conditional's reverse pass (M = 136: 9 row fragments, 7 idle waves)
        with ctx.options(fused_bwd_min_cols=min_cols):
            e, grads = model.compute_gradients(X, Y, zs=zs)
        assert abs(e - eo) <= RTOL * abs(eo)
        for li, (g, o) in enumerate(zip(grads, go)):
            for name, val in o.items():
                err = np.abs(g[name] - val).max()
                assert err < 1e-7 * np.abs(val).max() or err < 1e-8, (min_cols, li, name, err, np.abs(val).max())
    model.close()


def test_adam_bias_correction_uses_the_models_own_step_count(ctx):
    """adam_step() without t counts the steps taken on this model's (zero-initialised) moment buffers: identical to explicit
    t = 1, 2, ... and independent of whatever global_step a resumed run carries (a fresh tf optimiser restarts its beta powers)."""
    hwc, N = (12, 12, 1), 3
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 8, S=2, num_data=100, seed=4, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, N, seed=4)
    zs = syn.make_noise(spec, N, seed=4)
    a, b = build_from_spec(spec, X, Y), build_from_spec(spec, X, Y)
    for t in (1, 2, 3):
        a.compute_gradients(X, Y, zs=zs, fetch=False); a.adam_step(0.01, t)
        b.compute_gradients(X, Y, zs=zs, fetch=False); b.adam_step(0.01)
    a.pull_parameters(); b.pull_parameters()
    for la, lb in zip(a.layers, b.layers):
        np.testing.assert_array_equal(la.q_mu, lb.q_mu)
        np.testing.assert_array_equal(la.feature.Z, lb.feature.Z)
    a.close(); b.close()


def test_training_entry_points_fail_loudly(ctx):
    """Error behaviour of the training-step C-ABI: no silent fallbacks."""
    from deepcgp_amd import device as dev
    hwc, N = (12, 12, 1), 2
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 8, S=2, num_data=100, seed=2, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, N, seed=2)
    model = build_from_spec(spec, X, Y)
    with pytest.raises(dev.DcgpError):            # optimiser step before any gradient exists
        model._build(); model.adam_step(0.01, 1)
    model.compute_gradients(X, Y, fetch=False)
    with pytest.raises(dev.DcgpError):            # t is 1-based (None / 0 = the model's own step count)
        model.adam_step(0.01, -1)
    with pytest.raises(dev.DcgpError):            # learning rate must be positive
        model.adam_step(-1.0, 1)
    L = dev.lib()
    buf = np.zeros(3)
    assert L.dcgp_model_get_grad(model._model, 0, b"nope", buf.ctypes.data, 3) != 0
    assert L.dcgp_model_get_grad(model._model, 0, b"q_mu", buf.ctypes.data, 3) != 0      # wrong count
    assert L.dcgp_model_get_grad(model._model, 0, b"w", buf.ctypes.data, 3) != 0         # conv layers have no patch weights
    assert L.dcgp_model_get_param(model._model, 7, b"Z", buf.ctypes.data, 3) != 0        # no such layer
    with pytest.raises(dev.DcgpError):            # the one-call step checks the optimiser's arguments before anything is enqueued
        model.train_step(X, Y, -1.0)
    model.close()
    # a training step whose K_uu is not positive definite (two identical inducing patches, a variance that swallows the jitter) raises --
    # and the update, already enqueued behind the reverse pass, has left every parameter and the step count where they were
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 8, S=2, num_data=100, seed=2, conv_q_sqrt_scale=0.3, variance=1e15)
    spec["convs"][0]["Z"] = np.array(spec["convs"][0]["Z"])
    spec["convs"][0]["Z"][1] = spec["convs"][0]["Z"][0]
    bad = build_from_spec(spec, X, Y)
    bad._build()
    bad.pull_parameters()
    before = [(np.array(l.feature.Z), np.array(l.q_mu), np.array(l.q_sqrt)) for l in bad.layers]
    with pytest.raises(dev.DcgpError):
        bad.train_step(X, Y, 0.05)
    bad.pull_parameters()
    for (z0, m0, s0), l in zip(before, bad.layers):
        assert np.array_equal(l.feature.Z, z0) and np.array_equal(l.q_mu, m0) and np.array_equal(l.q_sqrt, s0)
    bad.close()


def test_gradient_properties_at_full_baseline_size(ctx):
    """BASELINE configs[1] at full size (M = 256, batch 32, S = 10: 46080 patch columns), where the oracle is too slow to
    be the checker: (1) the shard gradients of a 2-way batch split add up to the full-batch gradient, (2) the gradient
    agrees with central differences of the DEVICE forward ELBO along a random direction of every parameter group."""
    cfg = syn.CONFIGS["cfg2_mnist_CH_M256"]
    S, N = 10, cfg["batch"]
    spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=S, num_data=cfg["num_data"], seed=31, conv_q_sqrt_scale=0.1)
    X, Y = syn.make_batch(cfg["hwc"], N, seed=31)
    zs = syn.make_noise(spec, N, seed=31)
    model = build_from_spec(spec, X, Y)
    scale = cfg["num_data"] / N
    e, g = model.compute_gradients(X, Y, zs=zs, scale=scale, shards=1)
    assert abs(e - model.compute_log_likelihood(X, Y, zs=zs, scale=scale)) <= 1e-10 * abs(e)
    h = N // 2
    parts = [model.compute_gradients(X[lo:hi], Y[lo:hi], zs=[z[:, lo:hi] for z in zs], scale=scale, shards=2)
             for lo, hi in ((0, h), (h, N))]
    for li, gl in enumerate(g):
        for name, val in gl.items():
            tot = parts[0][1][li][name] + parts[1][1][li][name]
            assert np.abs(tot - val).max() <= 1e-8 * max(np.abs(val).max(), 1.0), (li, name)
    model.compute_gradients(X, Y, zs=zs, scale=scale, shards=1, fetch=False)     # restore the 1-shard KL weight
    rng = np.random.default_rng(0)
    for li, l in enumerate(model.layers):
        head = li == len(model.layers) - 1
        kern = l.kern.base_kernel if head else l.base_kernel
        for name in ("q_mu", "Z", "variance", "lengthscales"):
            if name == "q_mu":
                v0 = np.array(l.q_mu); set_ = lambda v, l=l: setattr(l, "q_mu", v)
            elif name == "Z":
                v0 = np.array(l.feature.Z); set_ = lambda v, l=l: setattr(l.feature, "Z", v)
            else:
                v0 = np.array(getattr(kern, name), np.float64); set_ = lambda v, k=kern, n=name: setattr(k, n, float(v))
            d = rng.standard_normal(v0.shape)
            hstep = 1e-6 * max(np.abs(v0).max(), 1.0)
            vals = []
            for sgn in (1.0, -1.0):
                set_(v0 + sgn * hstep * d)
                model.sync_parameters()
                vals.append(model.compute_log_likelihood(X, Y, zs=zs, scale=scale))
            set_(v0)
            model.sync_parameters()
            fd = (vals[0] - vals[1]) / (2 * hstep)
            an = float(np.sum(g[li][name] * d))
            assert abs(fd - an) <= 2e-4 * max(abs(fd), abs(an), 1.0), (li, name, fd, an)
    model.close()


def test_sgd_natgrad_and_trainable_flags(ctx):
    """(1) One SGD step equals theta + lr * (oracle gradient) in the unconstrained space.  (2) set_trainable(False)
    parameters are left alone by the device steps.  (3) Natural gradients: with the data term switched off (scale = 0)
    the objective is -KL[q || prior], conjugate in q, so ONE natural-gradient step with gamma = 1 must land on the prior
    (q_mu = 0, q_sqrt q_sqrt^T = K_uu(Z0) / K_uu(Z) / I) whatever q was -- the property that pins the step without gpflow.
    (4) The NatGrad + Adam loop of experiment.py:90-107 runs and improves the bound."""
    from oracle.grad import elbo_and_grad
    from deepcgp_amd.models import train
    hwc, N, S = (12, 12, 1), 3, 2
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 10, S=S, num_data=300, seed=8, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5)
    X, Y = syn.make_batch(hwc, N, seed=8)
    zs = syn.make_noise(spec, N, seed=8)
    ref = oracle_model(spec, X, Y)
    model = build_from_spec(spec, X, Y)
    # (1) + (2)
    model.set_trainable(0, "Z", False)
    model.compute_gradients(X, Y, zs=zs, fetch=False)
    model.sgd_step(1e-3)
    _, go = elbo_and_grad(ref, X, Y, zs)
    Z0_before = np.array(model.layers[0].feature.Z)
    qmu_before = [np.array(l.q_mu) for l in model.layers]
    var_before = model.layers[0].base_kernel.variance
    model.pull_parameters()
    assert np.array_equal(model.layers[0].feature.Z, Z0_before)                       # frozen
    for li, l in enumerate(model.layers):
        assert rel(l.q_mu, qmu_before[li] + 1e-3 * go[li]["q_mu"]) < 1e-9
    u = np.log(np.expm1(var_before - 1e-6)) + 1e-3 * go[0]["variance"] * (1.0 - np.exp(-(var_before - 1e-6)))
    assert abs(model.layers[0].base_kernel.variance - (np.log1p(np.exp(u)) + 1e-6)) < 1e-9
    model.set_trainable(0, "Z", True)
    # (3)
    model.compute_gradients(X, Y, zs=zs, scale=0.0, fetch=False)
    model.natgrad_step(1.0)
    model.pull_parameters()
    for li, l in enumerate(model.layers):
        o = ref.layers[li]
        head = li == len(model.layers) - 1
        K = (o.kern.base_kernel.K(l.feature.Z) if head else o.base_kernel.K(np.asarray(spec["convs"][li]["Z0"]))) + 1e-3 * np.eye(l.num_inducing)
        # hyper-parameters moved in step (1): evaluate the prior with the model's current values
        kern = l.kern.base_kernel if head else l.base_kernel
        Zp = l.feature.Z if head else np.asarray(spec["convs"][li]["Z0"])
        d2 = ((Zp[:, None, :] - Zp[None, :, :]) ** 2).sum(-1)
        K = kern.variance * np.exp(-0.5 * d2 / kern.lengthscales ** 2) + 1e-3 * np.eye(l.num_inducing)
        assert np.abs(l.q_mu).max() < 1e-6
        for r in range(l.q_sqrt.shape[0]):
            assert rel(l.q_sqrt[r] @ l.q_sqrt[r].T, K) < 1e-6
    # (3b) a step on the real objective against the NumPy restatement of the algorithm (tests/natgrad_ref.py), and the
    # positive-definiteness failure: a far too large gamma must raise and leave the parameters untouched
    from natgrad_ref import natgrad_reference
    _, g = model.compute_gradients(X, Y, zs=zs)
    before = [(np.array(l.q_mu), np.array(l.q_sqrt)) for l in model.layers]
    model.natgrad_step(0.01)
    model.pull_parameters()
    for (mu0, L0), gl, l in zip(before, g, model.layers):
        mu1, L1 = natgrad_reference(mu0, L0, gl["q_mu"], gl["q_sqrt"], 0.01)
        assert rel(l.q_mu, mu1) < 1e-8 and rel(l.q_sqrt, L1) < 1e-8
    model.compute_gradients(X, Y, zs=zs, fetch=False)
    kept = [(np.array(l.q_mu), np.array(l.q_sqrt)) for l in model.layers]
    with pytest.raises(np.linalg.LinAlgError):
        model.natgrad_step(1e6)
    model.pull_parameters()
    for (mu0, L0), l in zip(kept, model.layers):
        assert np.array_equal(l.q_mu, mu0) and np.array_equal(l.q_sqrt, L0)
    # (4)
    model.minibatch_size = N
    hist = train(model, 15, lr=0.01, optimizer="NatGrad", gamma=0.05, seed=3)
    assert len(hist) == 15 and np.mean(hist[-4:]) > np.mean(hist[:4])
    with pytest.raises(ValueError):
        train(model, 1, optimizer="LBFGS")
    model.close()


def test_gradient_with_device_rng_matches_explicit_noise(ctx):
    """The training loop draws its noise on the device (Philox stream keyed by the seed); the reverse pass recovers it
    from the stored sample.  Feeding the recovered noise explicitly must give the same ELBO and the same gradient."""
    from oracle.gpflow_ref import JITTER
    hwc, N, S = (14, 14, 1), 4, 3
    spec = syn.make_spec(hwc, [(3, 1, 3), (4, 2, 2)], (3, 1), 20, S=S, num_data=500, seed=12, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5)
    X, Y = syn.make_batch(hwc, N, seed=12)
    model = build_from_spec(spec, X, Y)
    e1, g1 = model.compute_gradients(X, Y, seed=77)
    Fs, Fm, Fv = model.propagate(X, S=S, seed=77)
    zs = [(f - m) / np.sqrt(v + JITTER) for f, m, v in zip(Fs[:-1], Fm[:-1], Fv[:-1])] + [None]
    e2, g2 = model.compute_gradients(X, Y, zs=zs)
    assert abs(e1 - e2) <= 1e-9 * abs(e1)
    for a, b in zip(g1, g2):
        for name in a:
            err = np.abs(a[name] - b[name]).max()
            assert err <= 1e-7 * max(np.abs(a[name]).max(), 1.0), (name, err)
    model.close()


def test_gradient_with_layer0_dedup_is_identical(ctx):
    """dedup_layer0 in the training step: the first layer's conditional and its reverse pass on the N distinct images
    (the S per-sample gradients of an image added first) give the gradients of the tiled evaluation."""
    hwc, N, S = (14, 14, 1), 4, 3
    spec = syn.make_spec(hwc, [(3, 1, 3), (4, 2, 2)], (3, 1), 20, S=S, num_data=500, seed=14, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5)
    X, Y = syn.make_batch(hwc, N, seed=14)
    zs = syn.make_noise(spec, N, seed=14)
    model = build_from_spec(spec, X, Y)
    e0, g0 = model.compute_gradients(X, Y, zs=zs)
    model.dedup_layer0 = True
    e1, g1 = model.compute_gradients(X, Y, zs=zs)
    assert abs(e0 - e1) <= 1e-12 * abs(e0)
    for a, b in zip(g0, g1):
        for name in a:
            err = np.abs(a[name] - b[name]).max()
            assert err <= 1e-9 * max(np.abs(a[name]).max(), 1.0), (name, err)
    model.close()


def test_gradients_are_bitwise_reproducible_at_full_size(ctx):
    """No atomics, fixed-order split-k reductions, stream joins where buffers are shared: repeated evaluations of the
    training step at BASELINE configs[1] give bit-identical ELBO and gradients (tiled and de-duplicated paths)."""
    cfg = syn.CONFIGS["cfg2_mnist_CH_M256"]
    spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=10, num_data=cfg["num_data"], seed=1, conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(cfg["hwc"], cfg["batch"], seed=1)
    model = build_from_spec(spec, X, Y)
    first = {}
    for rep in range(8):
        model.dedup_layer0 = bool(rep % 2)
        e, g = model.compute_gradients(X, Y, seed=7)
        flat = np.concatenate([np.ravel(v) for gl in g for v in gl.values()])
        if model.dedup_layer0 not in first:
            first[model.dedup_layer0] = (e, flat)
        else:
            assert e == first[model.dedup_layer0][0] and np.array_equal(flat, first[model.dedup_layer0][1]), rep
    # and the two paths agree with each other (not bitwise: different summation orders)
    assert abs(first[True][0] - first[False][0]) <= 1e-10 * abs(first[False][0])
    assert np.abs(first[True][1] - first[False][1]).max() <= 1e-8 * max(np.abs(first[False][1]).max(), 1.0)
    # the conditional's column-wise reverse pass as ONE strip-resident launch (default at this size, csrc/conv_bwd_fused.hip) against
    # its launch-per-product form at the full 46 080 columns: the same gradients to rounding
    import os
    with ctx.options(no_fused_bwd=1):
        for dedup in (False, True):
            model.dedup_layer0 = dedup
            e, g = model.compute_gradients(X, Y, seed=7)
            flat = np.concatenate([np.ravel(v) for gl in g for v in gl.values()])
            assert abs(e - first[dedup][0]) <= 1e-12 * abs(e)
            assert np.abs(flat - first[dedup][1]).max() <= 1e-9 * max(np.abs(flat).max(), 1.0), dedup
    model.close()


def test_model_destroy_releases_its_workspaces(ctx):
    """The forward / reverse-pass workspaces of a model live in the ctx under the model's prefix; destroying the model frees
    them (the training step's are large), so building models in a loop does not grow device memory."""
    from deepcgp_amd import device as dev
    import ctypes as C
    hwc, N = (28, 28, 1), 8
    spec = syn.make_spec(hwc, [(5, 2, 10)], (5, 1), 64, S=4, num_data=1000, seed=3, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, N, seed=3)

    def free_bytes():
        import subprocess  # noqa: F401  (hipMemGetInfo through the runtime the library already loaded)
        hip = C.CDLL("libamdhip64.so")
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    build_from_spec(spec, X, Y).close()             # first-use allocations of the ctx itself (shared scratch) happen here
    m = build_from_spec(spec, X, Y)
    m.compute_gradients(X, Y, fetch=False)
    m.close()
    def three_models():
        for _ in range(3):
            m = build_from_spec(spec, X, Y)
            m.compute_gradients(X, Y, fetch=False)
            m.close()
        return free_bytes()
    base = free_bytes()
    f1 = three_models()
    f2 = three_models()
    # a leak grows with every model (a training step's workspaces are tens of MB); a one-off allocation of the runtime's (a code object loaded late,
    # a pool grown once: seen once in a dozen full runs as 8+ MB between `base` and the first three models) does not
    assert f1 - f2 < 8 * 1024 * 1024, (base, f1, f2)
    assert base - f2 < 64 * 1024 * 1024, (base, f1, f2)


@pytest.mark.parametrize("maps,strip_kernel", [(17, False), (13, True), (1, True)])
def test_gradients_match_oracle_odd_shapes(ctx, maps, strip_kernel):
    """Padding paths of the reverse pass: M = 33 (Mp = 48), 17 feature maps (R padded to 32), two input channels,
    stride 3, a 2 x 2 head filter, 5 images x 3 samples.  With 13 maps / 1 map the conditional's column-wise adjoint also runs as
    the one-launch strip kernel (R <= 16; 240 columns = three full strips and a ragged one, 3 of 16 waves with rows)."""
    import os
    from oracle.grad import elbo_and_grad
    hwc, N, S = (13, 13, 2), 5, 3
    spec = syn.make_spec(hwc, [(4, 3, maps)], (2, 1), 33, S=S, num_data=777, seed=41, conv_q_sqrt_scale=0.3, variance=1.3, ls=2.1)
    rng = np.random.default_rng(41)
    spec["head"]["w"] = 0.5 + rng.random(spec["head"]["w"].shape)
    X, Y = syn.make_batch(hwc, N, seed=41)
    zs = syn.make_noise(spec, N, seed=41)
    ref = oracle_model(spec, X, Y)
    model = build_from_spec(spec, X, Y)
    eo, go = elbo_and_grad(ref, X, Y, zs)
    with ctx.options(fused_bwd_min_cols=0 if strip_kernel else -1):
        for dedup in (False, True):
            model.dedup_layer0 = dedup
            e, grads = model.compute_gradients(X, Y, zs=zs)
            assert abs(e - eo) <= RTOL * abs(eo)
            for li, (g, o) in enumerate(zip(grads, go)):
                for name, val in o.items():
                    err = np.abs(g[name] - val).max()
                    assert err < 1e-7 * np.abs(val).max() or err < 1e-8, (dedup, li, name, err, np.abs(val).max())
    model.close()


def test_no_read_of_unwritten_workspace_memory(ctx):
    """DCGP_POISON_WS=1 fills every fresh workspace with NaNs (csrc/ctx.hip): a kernel that reads a padded row / column it was
    supposed to receive initialised (M not a multiple of 16, ragged column counts) then poisons the result.  Run a forward
    ELBO, a prediction and two training steps (tiled, de-duplicated) of such a model in a fresh process and compare with
    this process's values."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from deepcgp_amd import synthetic as syn
from deepcgp_amd.models import build_from_spec
hwc, N, S = (13, 13, 2), 5, 3
spec = syn.make_spec(hwc, [(4, 3, 7), (3, 1, 3)], (2, 1), 21, S=S, num_data=777, seed=43, conv_q_sqrt_scale=0.3, variance=1.3, ls=2.1)
X, Y = syn.make_batch(hwc, N, seed=43)
zs = syn.make_noise(spec, N, seed=43)
model = build_from_spec(spec, X, Y)
out = [model.compute_log_likelihood(X, Y, zs=zs)]
out.append(float(np.sum(model.predict_y(X, S, zs=zs)[0])))
for dedup in (False, True):
    model.dedup_layer0 = dedup
    e, g = model.compute_gradients(X, Y, zs=zs)
    out.append(e)
    out.append(float(sum(np.sum(np.abs(v)) for gl in g for v in gl.values())))
print("RESULT", " ".join(repr(v) for v in out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ)
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
        return [float(v) for v in line.split()[1:]]
    clean, poisoned = run({}), run({"DCGP_POISON_WS": "1"})
    assert all(np.isfinite(poisoned)), poisoned
    assert clean == poisoned, (clean, poisoned)


def test_prologues_ahead_with_fewer_cus_than_workgroups(ctx):
    """The layer kernel's prologues ahead (csrc/conv_fused.hip) hand A1 from one workgroup of a launch to another.  On a CU-masked main stream
    (DCGP_CU_PARTITION=1: steps in flight run their data path on 240 of the 256 CUs) the launch has more workgroups than CUs: some start only when
    others have left, and none of them may hold an item a running workgroup waits for -- every item, the first included, comes off the device
    counter.  Same ELBO as the synchronous step on the whole chip, to the last bit, in a fresh process (the switch is read at ctx creation)."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from deepcgp_amd import synthetic as syn
from deepcgp_amd.models import build_from_spec
spec, X, Y = syn.make_config("cfg2_mnist_CH_M256")
model = build_from_spec(spec, X, Y)
out = [model.compute_log_likelihood(X, Y, seed=5)]
tickets = [model.enqueue_log_likelihood(X, Y, seed=5) for _ in range(3)]
out += [model.collect_log_likelihood(t) for t in tickets]
print("RESULT", " ".join(repr(v) for v in out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ)
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
        return [float(v) for v in line.split()[1:]]
    whole, masked, off = run({}), run({"DCGP_CU_PARTITION": "1"}), run({"DCGP_CU_PARTITION": "1", "DCGP_FUSED_PRE": "0"})
    assert all(np.isfinite(masked)), masked
    assert len(set(whole)) == 1 and whole == masked == off, (whole, masked, off)


def test_one_launch_factorisation_chain_matches_the_launch_per_panel_chain(ctx):
    """DCGP_CHOL_ONE_LAUNCH=1 runs the Cholesky + inverse chain (conv_gp/conditionals.py:29, layers.py:151,156) as ONE launch whose
    workgroups hand panels to each other through flags (csrc/chol_fused.hip, chol_persist_kernel; opt-in: measured slower than the
    launches it replaces).  Same arithmetic per tile: the ELBO and a training step of a 3-layer model with a ragged M (41: a 16-wide
    last panel) and of an M = 256 head must agree with the default chain to rounding, poisoned workspaces included."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from deepcgp_amd import synthetic as syn
from deepcgp_amd.models import build_from_spec
out = []
for hwc, convs, head, M in (((13, 13, 2), [(4, 3, 7), (3, 1, 3)], (2, 1), 41), ((28, 28, 1), [], (5, 1), 256)):
    N, S = 4, 2
    spec = syn.make_spec(hwc, convs, head, M, S=S, num_data=777, seed=47, conv_q_sqrt_scale=0.3, variance=1.3, ls=2.1)
    X, Y = syn.make_batch(hwc, N, seed=47)
    zs = syn.make_noise(spec, N, seed=47)
    model = build_from_spec(spec, X, Y)
    for rep in range(3):          # the flags are monotone across launches: several steps on the same sync area
        out.append(model.compute_log_likelihood(X, Y, zs=zs))
    e, g = model.compute_gradients(X, Y, zs=zs)
    out.append(e)
    out.append(float(sum(np.sum(np.abs(v)) for gl in g for v in gl.values())))
    model.close()
print("RESULT", " ".join(repr(v) for v in out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ)
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
        return np.array([float(v) for v in line.split()[1:]])
    ref, one = run({}), run({"DCGP_CHOL_ONE_LAUNCH": "1", "DCGP_POISON_WS": "1"})
    assert np.all(np.isfinite(one)), one
    assert np.max(np.abs(one - ref) / np.maximum(np.abs(ref), 1.0)) < 1e-9, (ref, one)


def test_two_gpu_bench_runs_rccl_and_matches_one_rank(ctx):
    """Skipped unless the box has two GPUs (the round's 1-GPU boxes skip it; the driver's 8-GPU node runs it): `bench.py --gpus 2`
    must come up over RCCL -- both ranks in one communicator, not the host fallback -- and reproduce the 1-rank ELBO of the same
    global batch (image-sharded data term + one in-stream ncclAllReduce)."""
    import json
    import subprocess
    import sys
    from deepcgp_amd import device as dev
    if dev.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")

    def line(gpus):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                            "--no-grad-leg", "--no-extra-legs"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    one, two = line(1), line(2)
    assert two["n_gpus"] == 2 and two["ranks_seen_by_rccl"] == 2, two["config"]
    assert "ncclAllReduce" in two["config"]["parallelism"]
    assert abs(two["elbo"] - one["elbo"]) <= 1e-9 * abs(one["elbo"]), (one["elbo"], two["elbo"])
    assert two["steps_per_s_two_in_flight"] is not None


def test_two_gpu_rccl_exchange_modes_agree(ctx):
    """Skipped unless the box has two GPUs.  The training step's two gradient exchanges over a REAL communicator (ncclAllReduce against
    ncclReduceScatter -> Adam on the rank's shard -> ncclAllGather, in place on the padded gradient block and the staging block): the same
    parameters on both ranks and, to rounding of the reduction order, in both modes; a switch back to mode 0 or a full Adam step on top of
    rank-local moments is refused (tests/rccl_exchange_worker.py; one GPU: the virtual-rank replay of test_sharded_adam_step_equals_the_full_step)."""
    import subprocess
    import sys
    from deepcgp_amd import device as dev
    from deepcgp_amd.dist import spawn_ranks
    if dev.device_count() < 2:
        pytest.skip("needs two GPUs")
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)

    def run(mode):
        code = ("import sys; sys.path[:0] = [%r, %r]; from deepcgp_amd.dist import spawn_ranks; "
                "sys.exit(spawn_ranks(2, [%r, %r]))" % (root, here, os.path.join(here, "rccl_exchange_worker.py"), str(mode)))
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, here]), HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        out = {ln.split()[0]: np.array([float(v) for v in ln.split()[1:]]) for ln in r.stdout.splitlines() if ln.startswith(("PARAMS", "ELBOS"))}
        return out["PARAMS"], out["ELBOS"]
    p0, e0 = run(0)
    p1, e1 = run(1)
    assert np.all(np.isfinite(p0)) and np.all(np.isfinite(p1))
    assert np.max(np.abs(e1 - e0) / np.abs(e0)) < 1e-9, (e0, e1)
    assert np.max(np.abs(p1 - p0)) < 1e-9 * max(1.0, np.max(np.abs(p0))), np.max(np.abs(p1 - p0))


@pytest.mark.parametrize("variant", ["head", "conv"])
def test_learns_real_digits(ctx, variant):
    """End-to-end learning evidence on REAL images (the reference's only published kind of number is accuracy,
    results/*/log.csv:16): sklearn.datasets.load_digits (8 x 8 digits, ships offline), the reference's flags through ModelBuilder,
    models.train (Adam branch of conv_gp/experiment.py:84-108), AccuracyLogger (conv_gp/utils/log.py:50-67).  The paper's "1-layer"
    (SVGP head with the ConvKernel) and one ConvLayer + head must both reach >= 0.93 test accuracy within 750 steps from chance,
    with the ELBO rising block over block (tools/digits_train.py prints the whole trajectory: 0.97 / 0.99 after 500 steps)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from digits_train import run
    out, _ = run(variant, 750)
    steps, elbos, accs = zip(*out)
    assert accs[0] < 0.3, accs                     # untrained: chance level (0.1)
    assert accs[-1] >= 0.93, accs
    assert elbos[1] < elbos[2] < elbos[3], elbos   # mean minibatch ELBO of steps 1-250 < 251-500 < 501-750


def _all_params(model):
    model.pull_parameters()
    out = []
    for li, l in enumerate(model.layers):
        head = li == len(model.layers) - 1
        kern = (l.kern.base_kernel if hasattr(l.kern, "base_kernel") else l.kern) if head else l.base_kernel
        out += [np.array(l.feature.Z), np.array(l.q_mu), np.array(l.q_sqrt), np.array(kern.variance), np.array(kern.lengthscales)]
        if head and hasattr(l.kern, "patch_weights"):
            out.append(np.array(l.kern.patch_weights))
    return out


@pytest.mark.parametrize("head_kernel", ["conv", "rbf"])
@pytest.mark.parametrize("ranks", [1, 2, 3, 8])
def test_sharded_adam_step_equals_the_full_step(ctx, head_kernel, ranks):
    """Exchange mode 1 of the multi-rank training step (dcgp_model_set_grad_exchange; SURVEY section 5: reduce-scatter -> Adam on the rank's
    shard -> all-gather of the parameters), its device part played on one GPU: dcgp_model_debug_sharded_adam updates shard 0 of every
    layer's parameter block in place and takes the other `ranks - 1` shards through the staging block and the unstage pass -- the block cut
    as dcgp_shard_range cuts it, across group boundaries, with a frozen group passing through -- and must land on exactly the parameters
    two plain dcgp_model_adam_step calls give (conv_gp/experiment.py:104-107's optimiser)."""
    hwc, N = (12, 12, 1), 3
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 11, S=2, num_data=200, seed=4, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5, head_kernel=head_kernel)
    X, Y = syn.make_batch(hwc, N, seed=4)
    res = []
    for sharded in (False, True):
        model = build_from_spec(spec, X, Y)
        model.set_trainable(0, "q_mu", False)          # a frozen group in the middle of layer 0's block
        for t in (1, 2):
            model.compute_gradients(X, Y, zs=syn.make_noise(spec, N, seed=50 + t), fetch=False)
            if sharded:
                model.debug_sharded_adam(ranks, 0.05, t)
            else:
                model.adam_step(0.05, t)
        res.append(_all_params(model))
        e_after = model.compute_log_likelihood(X, Y, zs=syn.make_noise(spec, N, seed=60))
        res[-1].append(np.array(e_after))
        model.close()
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


def test_rccl_single_rank_reduce_scatter_path(ctx):
    """Exchange mode 1 under a 1-rank RCCL communicator: the training step is the plain one (one rank owns every shard), and switching
    the mode back and forth changes nothing."""
    from deepcgp_amd import device as dev
    hwc = (12, 12, 1)
    spec = syn.make_spec(hwc, [(3, 2, 4)], (3, 1), M=10, S=2, num_data=500, seed=5, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, 4, seed=5)
    outs = []
    for mode in (0, 1):
        model = build_from_spec(spec, X, Y)
        ctx.comm_init(1, 0, dev.comm_unique_id())
        try:
            model.set_grad_exchange(mode)
            es = [model.train_step(X, Y, 0.03, zs=syn.make_noise(spec, 4, seed=70 + t), t=t) for t in (1, 2, 3)]
        finally:
            dev.lib().dcgp_comm_destroy(ctx.handle)
        outs.append(_all_params(model) + [np.array(es)])
        model.close()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_allreduce_of_a_step_in_flight_does_not_hold_up_the_next_step(ctx):
    """SURVEY 8(e) / the review's item 6c, on one GPU with a 1-rank RCCL communicator: with steps kept in flight the data term's
    ncclAllReduce + ELBO assembly of step i run on the comm stream, so step i + 1's kernels on the main stream do not queue behind the
    collective.  Causal check: the comm stream is GATED (dcgp_debug_comm_gate) -- step 0's all-reduce cannot run -- two steps are enqueued,
    and the main stream must drain all the same (both data paths done) while neither result is there; the gate opens, both results
    arrive and equal the ungated ones.  With the collective in the main stream (ctx option comm_inline) the same schedule cannot drain."""
    import ctypes as C
    import time
    from deepcgp_amd import device as dev
    hwc = (12, 12, 1)
    spec = syn.make_spec(hwc, [(3, 2, 4)], (3, 1), M=10, S=2, num_data=500, seed=5, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, 4, seed=5)
    zs = syn.make_noise(spec, 4, seed=5)
    model = build_from_spec(spec, X, Y)
    L = dev.lib()
    idle = C.c_int(0)

    def main_idle_within(seconds):
        t0 = time.time()
        while time.time() - t0 < seconds:
            ctx._check(L.dcgp_debug_comm_gate(ctx.handle, 1, C.byref(idle)))
            if idle.value & 1:
                return True
            time.sleep(0.01)
        return False
    ctx.comm_init(1, 0, dev.comm_unique_id())
    try:
        want = [model.collect_log_likelihood(model.enqueue_log_likelihood(X, Y, zs=zs, seed=s)) for s in (1, 2)]
        ctx._check(L.dcgp_debug_comm_gate(ctx.handle, 1, None))                    # close the gate
        tickets = [model.enqueue_log_likelihood(X, Y, zs=zs, seed=s) for s in (1, 2)]
        assert main_idle_within(2.0), "the main stream waits for the gated all-reduce: the collective is not on the comm stream"
        assert not (idle.value & 2)                                                # ... while the comm stream still sits at the gate: no result yet
        ctx._check(L.dcgp_debug_comm_gate(ctx.handle, 0, None))                    # open it
        assert [model.collect_log_likelihood(t) for t in tickets] == want
        with ctx.options(comm_inline=1):                                           # the A/B: collective in the main stream
            ctx._check(L.dcgp_debug_comm_gate(ctx.handle, 1, None))
            tickets = [model.enqueue_log_likelihood(X, Y, zs=zs, seed=s) for s in (1, 2)]
            assert main_idle_within(1.0) and (idle.value & 2)      # nothing went to the comm stream: the whole step drains by itself
            ctx._check(L.dcgp_debug_comm_gate(ctx.handle, 0, None))
            assert [model.collect_log_likelihood(t) for t in tickets] == want
    finally:
        L.dcgp_debug_comm_gate(ctx.handle, 0, None)
        L.dcgp_comm_destroy(ctx.handle)
    model.close()


@pytest.mark.parametrize("arch", ["conv_head", "head_only"])
def test_factor_reuse_at_unchanged_parameters(ctx, arch):
    """Evaluation sweeps at one parameter state (AccuracyLogger / LogLikelihoodLogger, conv_gp/utils/log.py:55-68): predict_y / propagate skip the
    parameter-only chain while no parameter was written (default), compute_log_likelihood only in mode 2; results are bit-identical to steps that
    run it, and every way of writing a parameter -- sync_parameters, an Adam step, a training step -- ends the reuse."""
    hwc, N, S = (28, 28, 1), 6, 3
    convs = [(5, 2, 10)] if arch == "conv_head" else []
    spec = syn.make_spec(hwc, convs, (5, 1), 48, S=S, num_data=1000, seed=21, conv_q_sqrt_scale=0.2)
    X, Y = syn.make_batch(hwc, N, seed=21)
    zs = syn.make_noise(spec, N, seed=22)
    ref = oracle_model(spec, X, Y)
    model = build_from_spec(spec, X, Y)
    with ctx.options(no_factor_reuse=1):
        p0, _ = model.predict_y(X, S, zs=zs)
        e0 = model.compute_log_likelihood(X, Y, zs=zs)
    k0 = model.chain_skips
    assert k0 == 0
    p1, _ = model.predict_y(X, S, zs=zs)          # runs the chain (the ELBO step before it recorded one with KL pieces: another kind)
    p2, _ = model.predict_y(X, S, zs=zs)
    _, Fm, Fv = model.propagate(X, S=S, zs=zs)
    assert model.chain_skips == k0 + 2             # the second predict_y and propagate
    assert np.array_equal(p1, p0) and np.array_equal(p2, p0)
    _, om, ov = ref.propagate(X, S=S, zs=zs)
    assert rel(Fm[-1], om[-1]) < RTOL and rel(Fv[-1], ov[-1]) < RTOL
    e1 = model.compute_log_likelihood(X, Y, zs=zs)  # mode 1: the ELBO step always runs the chain
    assert model.chain_skips == k0 + 2 and e1 == e0
    model.set_factor_reuse(2)
    e2 = model.compute_log_likelihood(X, Y, zs=zs)  # same kind of chain as the step before it: reused
    e3 = model.compute_log_likelihood(X[:5], Y[:5], zs=[z[:, :5] for z in zs])   # another batch, same parameters: reused
    assert model.chain_skips == k0 + 4 and e2 == e0
    assert abs(e3 - ref.compute_log_likelihood(X[:5], Y[:5], zs=[z[:, :5] for z in zs])) <= RTOL * abs(e3)
    # a pushed parameter ends it, and the new value is what the next step sees
    model.layers[-1].q_mu = model.layers[-1].q_mu + 0.25
    ref.layers[-1].q_mu = ref.layers[-1].q_mu + 0.25
    model.sync_parameters()
    k1 = model.chain_skips
    e4 = model.compute_log_likelihood(X, Y, zs=zs)
    assert model.chain_skips == k1 and e4 != e0
    assert abs(e4 - ref.compute_log_likelihood(X, Y, zs=zs)) <= RTOL * abs(e4)
    e5 = model.compute_log_likelihood(X, Y, zs=zs)
    assert model.chain_skips == k1 + 1 and e5 == e4
    # an optimiser step ends it too; a training step's own forward pass never reuses
    model.compute_gradients(X, Y, zs=zs)
    k2 = model.chain_skips
    model.adam_step(1e-3)
    e6 = model.compute_log_likelihood(X, Y, zs=zs)
    assert model.chain_skips == k2 and e6 != e4
    model.train_step(X, Y, 1e-3, zs=zs)
    model.train_step(X, Y, 1e-3, zs=zs)
    assert model.chain_skips == k2
    model.set_factor_reuse(0)
    model.predict_y(X, S, zs=zs)
    model.predict_y(X, S, zs=zs)
    assert model.chain_skips == k2
    model.close()


def test_persistent_layer_launch_and_prep_placement_are_bit_identical(ctx):
    """ctx options of round 6: the conv layer kernel as a persistent launch (strips dealt by a device counter or by a fixed stride, one or two
    workgroups per CU) and the head-first model's operand preparation on either stream give the same ELBO to the last bit."""
    spec, X, Y = syn.make_config("cfg2_mnist_CH_M256")
    X, Y = X[:24], Y[:24]                       # 540 strips of 64 columns / 1080 of 32: more than one per slot of the chip
    model = build_from_spec(spec, X, Y)
    want = model.compute_log_likelihood(X, Y, seed=3)
    for kw in (dict(fused_persist=1), dict(fused_persist=2), dict(fused_shape=2, fused_persist=1, fused_stagger=0),
               dict(fused_shape=2, fused_persist=1, fused_stagger=25), dict(fused_shape=2, fused_persist=2),
               # prologues ahead (conv_fused.hip: phases 0 - 2 of later strips in the partial first round's spare workgroups): never, chosen, forced counts
               dict(fused_pre=0), dict(fused_pre=-1), dict(fused_pre=1), dict(fused_pre=3), dict(fused_pre=9)):
        with ctx.options(**kw):
            for rep in range(3):                # (the counters go back to zero behind every launch)
                assert model.compute_log_likelihood(X, Y, seed=3) == want, kw
    model.close()
    spec, X, Y = syn.make_config("cfg2_mnist_H_M256")
    model = build_from_spec(spec, X[:8], Y[:8])
    want = model.compute_log_likelihood(X[:8], Y[:8], seed=4)
    with ctx.options(prep_on_chain=1):
        assert model.compute_log_likelihood(X[:8], Y[:8], seed=4) == want
    model.close()
