"""CPU: host-side logic and the C-ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from deepcgp_amd import device as dev, synthetic as syn
from deepcgp_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = dev.lib()
    declared = dev.declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(dev._SIGS) == declared            # the ctypes table covers the whole header, nothing more


def test_no_device_fails_loudly():
    if dev.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(dev.DcgpError):
        dev.Context(0)


def test_null_ctx_is_an_argument_error():
    L = dev.lib()
    assert L.dcgp_sync(None) == dev.ERR_ARG
    assert L.dcgp_kuu_rbf(None, None, 4, 4, 1.0, 1.0, 0.0, None) == dev.ERR_ARG
    assert L.dcgp_model_destroy(None) == dev.ERR_ARG
    n = ctypes.c_int(-1)
    assert L.dcgp_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_fullview_host_geometry():
    from deepcgp_amd.views import FullView
    v = FullView((28, 28), 5, 1)
    assert (v.patch_count, v.patch_length, v.out_image_height, v.out_image_width) == (576, 25, 24, 24)
    v = FullView((28, 28), 5, 1, stride=2)
    assert (v.patch_count, v.out_image_height) == (144, 12)
    v = FullView((32, 32), 4, 3, stride=2)
    assert (v.patch_count, v.patch_length) == (225, 48)
    v = FullView((15, 15, 10), 5, 10)
    assert (v.patch_count, v.patch_length, v.dilation, v.patch_shape) == (121, 250, 1, [5, 5])
    with pytest.raises(ValueError):
        FullView((4, 4), 5, 1)


def test_parse_ints_and_flags():
    from deepcgp_amd.models import parse_ints
    from deepcgp_amd.arguments import default_parser, train_steps
    assert parse_ints('') == [] and parse_ints('384,384') == [384, 384]
    flags = default_parser().parse_args(['--name', 'x'])
    assert (flags.M, flags.feature_maps, flags.filter_sizes, flags.strides) == ('384,384', '10', '5,5', '2,1')
    assert flags.batch_size == 32 and flags.num_samples == 10 and not flags.white and flags.last_kernel == 'conv'
    assert train_steps(flags) == 5


def test_synthetic_configs_shapes():
    spec, X, Y = syn.make_config("cfg1_mnist_H_M32", S=2)
    assert X.shape == (32, 784) and Y.shape == (32,) and spec["head"]["Z"].shape == (32, 25) and not spec["convs"]
    spec = syn.make_spec((28, 28, 1), [(5, 2, 10)], (5, 1), M=8, S=2)
    c, h = spec["convs"][0], spec["head"]
    assert (c["H"], c["W"], c["C"]) == (28, 28, 1) and (h["H"], h["W"], h["C"]) == (12, 12, 10)
    assert h["Z"].shape == (8, 250) and h["w"].shape == (64,) and c["q_sqrt"].shape == (10, 8, 8)
    assert syn.layer_output_dims(spec) == [1440, 10]
    zs = syn.make_noise(spec, 3)
    assert [z.shape for z in zs] == [(2, 3, 1440), (2, 3, 10)]
    spec3 = syn.make_spec((32, 32, 3), [(4, 2, 10), (5, 1, 10)], (5, 1), M=4, S=1)
    assert [(c["H"], c["C"]) for c in spec3["convs"]] == [(32, 3), (15, 10)] and spec3["head"]["H"] == 11


def test_shard_range():
    for n in (0, 1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_two_rank_gloo_elbo_allreduce():
    """world_size-2 run of the N>1 assembly path on CPU: gloo all-reduce of the per-rank data term."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    worker = os.path.join(ROOT, "tests", "gloo_worker.py")
    procs = [subprocess.Popen([sys.executable, worker], env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    vals = [float(o.strip().splitlines()[-1].split()[-1]) for o in outs]
    assert vals[0] == vals[1]
    assert "OK" in outs[0]


def test_two_rank_hostgroup_elbo(tmp_path):
    """The same world_size-2 assembly over the product's own host group (TCP rendezvous through a file, no torch): id
    broadcast, sum / max all-reduce, barrier; both ranks arrive at the full-batch ELBO."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DCGP_RDZV_FILE=str(tmp_path / "rdzv"),
               PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    worker = os.path.join(ROOT, "tests", "hostgroup_worker.py")
    for world in (2, 3):
        procs = [subprocess.Popen([sys.executable, worker], env=dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = [p.communicate(timeout=300)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        vals = [float(o.strip().splitlines()[-1].split()[-1]) for o in outs]
        assert len(set(vals)) == 1 and all("OK" in o for o in outs)


def test_hostgroup_under_an_external_launcher():
    """The driver starts multi-GPU runs with `python -m torch.distributed.run ... bench.py --gpus N`: the launcher owns MASTER_PORT,
    so the host group meets through a file named after the launcher's pid + port and an ephemeral port of its own."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    env.pop("DCGP_RDZV_FILE", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29733", os.path.join(ROOT, "tests", "hostgroup_worker.py")],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK rank") == 2


def test_spawn_ranks_sets_the_rank_environment(tmp_path):
    """spawn_ranks: one child per rank with RANK / LOCAL_RANK / WORLD_SIZE and a shared rendezvous file; worst exit code."""
    from deepcgp_amd.dist import spawn_ranks
    script = tmp_path / "child.py"
    script.write_text("import os, sys\n"
                      "open(os.path.join(%r, 'r' + os.environ['RANK']), 'w').write(os.environ['WORLD_SIZE'] + ' ' + os.environ['DCGP_RDZV_FILE'])\n"
                      "sys.exit(3 if os.environ['RANK'] == '1' and len(sys.argv) > 1 else 0)\n" % str(tmp_path))
    assert spawn_ranks(2, [str(script)]) == 0
    got = [(tmp_path / ("r%d" % r)).read_text().split() for r in range(2)]
    assert got[0][0] == got[1][0] == "2" and got[0][1] == got[1][1]
    assert spawn_ranks(2, [str(script), "fail"]) == 3


def test_spawn_ranks_eight_ranks_through_the_host_group(tmp_path):
    """The shape of the driver's 8-GPU run on CPU: spawn_ranks(8) starts the ranks, they meet over the host group (file rendezvous,
    0600), ship the 128-byte id, sum the sharded data term of a batch of 11 images (ragged shards: 2,2,2,1,1,1,1,1) to the full-batch
    ELBO, and a rank that dies takes the others down with its exit code instead of leaving them in the group's socket timeout."""
    from deepcgp_amd.dist import spawn_ranks, shard_range
    assert [shard_range(11, r, 8)[1] - shard_range(11, r, 8)[0] for r in range(8)] == [2, 2, 2, 1, 1, 1, 1, 1]
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from deepcgp_amd import synthetic as syn\n"
        "from deepcgp_amd.dist import HostGroup, shard_batch, assemble_elbo, env_rank_world\n"
        "from oracle_build import oracle_model\n"
        "rank, world, _ = env_rank_world()\n"
        "if len(sys.argv) > 1 and rank == 5: sys.exit(7)          # dies before the rendezvous\n"
        "grp = HostGroup(rank, world, timeout=60.0)\n"
        "assert grp.broadcast_bytes(bytes(range(128)) if rank == 0 else b'') == bytes(range(128))\n"
        "hwc = (8, 8, 1)\n"
        "spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), M=5, S=2, num_data=500, seed=3, conv_q_sqrt_scale=0.3)\n"
        "X, Y = syn.make_batch(hwc, 11, seed=3); zs = syn.make_noise(spec, 11, seed=3)\n"
        "model = oracle_model(spec, X, Y)\n"
        "Xs, Ys, zl = shard_batch(X, Y, zs, rank, world)\n"
        "total = float(grp.allreduce([model.data_term(Xs, Ys, zs=zl)], 'sum')[0])\n"
        "elbo = assemble_elbo(total, model.KL(), spec['num_data'], 11)\n"
        "full = model.compute_log_likelihood(X, Y, zs=zs)\n"
        "assert abs(elbo - full) <= 1e-12 * abs(full), (elbo, full)\n"
        "assert float(grp.allreduce([float(rank)], 'max')[0]) == world - 1\n"
        "grp.barrier(); grp.close()\n"
        "open(os.path.join(%r, 'ok%%d' %% rank), 'w').write(repr(elbo))\n" % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path)))
    assert spawn_ranks(8, [str(script)]) == 0
    vals = {(tmp_path / ("ok%d" % r)).read_text() for r in range(8)}
    assert len(vals) == 1
    t0 = __import__("time").time()
    assert spawn_ranks(8, [str(script), "die"]) == 7
    assert __import__("time").time() - t0 < 50.0          # not the 60 s socket timeout of the survivors


def test_reference_format_checkpoint_fixture_and_loader(tmp_path):
    """tests/golden/checkpoint/ref_checkpoint_3layer.npy carries exactly the path names a reference-trained 3-layer model prints
    (notebooks/Inspect.ipynb cell 6) plus 'global_step' (conv_gp/experiment.py:56-64); the table-driven loader files every layer key
    under the right field, applies the reference's "last stored layer becomes the last model layer" rule and refuses a deeper file."""
    from golden.checkpoint.make_checkpoint_fixture import INSPECT_CELL6
    from deepcgp_amd.models import read_checkpoint, CHECKPOINT_FIELDS
    path = os.path.join(ROOT, "tests", "golden", "checkpoint", "ref_checkpoint_3layer.npy")
    raw = np.load(path, allow_pickle=True).item()
    assert set(raw) == set(INSPECT_CELL6) | {"global_step"}
    step, rec = read_checkpoint(path, 3)
    assert step == 25000 and sorted(rec) == [0, 1, 2]
    assert set(rec[0]) == set(rec[1]) == {"Z", "q_mu", "q_sqrt", "variance", "ls"} and set(rec[2]) == {"Z", "q_mu", "q_sqrt", "variance", "ls", "w"}
    assert rec[0]["Z"].shape == (6, 16) and rec[1]["Z"].shape == (6, 27) and rec[2]["Z"].shape == (6, 18) and rec[2]["w"].shape == (4,)
    assert rec[0]["q_sqrt"].shape == (3, 6, 6) and rec[1]["q_sqrt"].shape == (2, 6, 6) and rec[2]["q_sqrt"].shape == (10, 6, 6)
    for i in range(3):
        kern = "kern" if i == 2 else "conv_kernel"
        assert rec[i]["variance"] == raw["DGP/layers/%d/%s/base_kernel/variance" % (i, kern)]
        assert rec[i]["ls"] == raw["DGP/layers/%d/%s/base_kernel/lengthscales" % (i, kern)]
        np.testing.assert_array_equal(rec[i]["q_mu"], raw["DGP/layers/%d/q_mu" % i])
    # every key under layers/ is claimed by exactly the first matching row of the table
    for key in raw:
        if "/layers/" in key:
            assert sum(key.endswith(tail) for tail, _ in CHECKPOINT_FIELDS) >= 1, key
    _, rec4 = read_checkpoint(path, 4)            # a 4-layer model: the stored head parameters go to ITS last layer
    assert sorted(rec4) == [0, 1, 3] and "w" in rec4[3]
    with pytest.raises(AssertionError):
        read_checkpoint(path, 2)
    # the dense RBF head's keys (--last-kernel rbf): kernel directly under kern/
    dense = {"DGP/layers/0/kern/variance": np.array(2.0), "DGP/layers/0/kern/lengthscales": np.arange(1.0, 4.0),
             "DGP/layers/0/feature/Z": np.zeros((2, 3)), "global_step": 7}
    np.save(tmp_path / "dense.npy", dense)
    step, rec = read_checkpoint(str(tmp_path / "dense.npy"), 1)
    assert step == 7 and rec[0]["variance"] == 2.0 and rec[0]["ls"].shape == (3,) and rec[0]["Z"].shape == (2, 3)


def test_model_builder_stages_and_patch_draw():
    """flags -> stage list (conv_gp/arguments.py:27-31 comma lists, the two length invariants of models.py:54-55) and the
    vectorised patch draw: every row is one contiguous f x f window in FullView's (kh, kw, c) element order."""
    from deepcgp_amd.arguments import default_parser
    from deepcgp_amd.models import ModelBuilder, draw_patches
    fl = default_parser().parse_args(['--name', 't', '-M', '6,7,8', '--feature-maps', '3,2', '--filter-sizes', '4,3,3', '--strides', '2,1,1'])
    b = ModelBuilder(fl, np.zeros((4, 14, 14, 1)), np.zeros((4, 1)))
    assert b.stages() == ([(6, 4, 2, 3), (7, 3, 1, 2)], (8, 3, 1))
    fl.feature_maps = '3'
    with pytest.raises(AssertionError):
        b.stages()
    fl.feature_maps, fl.M = '', '32'                  # the paper's "1-layer": scalar M = head only (results/N60000_M256/options.toml)
    fl.filter_sizes, fl.strides = '5', '1'
    assert b.stages() == ([], (32, 5, 1))
    X = np.arange(3 * 6 * 5 * 2, dtype=np.float64).reshape(3, 6, 5, 2)
    rows = draw_patches(X, 50, 2, np.random.RandomState(1))
    assert rows.shape == (50, 8)
    for r in rows:
        w = r.reshape(2, 2, 2)
        assert w[0, 0, 1] - w[0, 0, 0] == 1 and w[0, 1, 0] - w[0, 0, 0] == 2 and w[1, 0, 0] - w[0, 0, 0] == 10
    assert len({tuple(r) for r in rows}) > 10


def test_flat_import_shim():
    """SURVEY 8(b) Level 1: the reference's modules are imported flat (``from layers import ConvLayer``, conv_gp/models.py:8-11);
    with deepcgp_amd/flat on sys.path the same statements resolve to this package's classes."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from layers import ConvLayer, MultiOutputConvKernel; from kernels import ConvKernel, PatchInducingFeatures, _sample_patches;"
            "from views import FullView; from conditionals import conditional; from models import ModelBuilder; from arguments import default_parser;"
            "from mean_functions import Conv2dMean, IdentityConv2dMean;"          # conv_gp/models.py:11
            "import deepcgp_amd.mean_functions as MF; assert Conv2dMean is MF.Conv2dMean and issubclass(Conv2dMean, IdentityConv2dMean);"
            "import deepcgp_amd.layers as L; assert ConvLayer is L.ConvLayer; print(FullView((28, 28), 5, 1).patch_count)"
            % (ROOT, os.path.join(ROOT, "deepcgp_amd", "flat")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "576", out.stderr


def test_mean_function_objects_without_a_device():
    """conv_gp/mean_functions.py:6-41 / models.py:95-100: constructor arguments, the two initial filters, set_trainable, and how
    ConvLayer classifies what it is handed (the kernels add Conv2dMean's centre pixel themselves only while the filter is the
    initial one and the geometry is the view's)."""
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.mean_functions import Conv2dMean, IdentityConv2dMean, Zero
    from deepcgp_amd.views import FullView
    idm = IdentityConv2dMean(5, 3, 4, stride=2)
    assert idm.conv_filter.shape == (5, 5, 3, 4) and idm.conv_filter.sum() == 12 and np.all(idm.conv_filter[2, 2] == 1.0)
    cm = Conv2dMean(5, 3, 4, stride=2)
    assert cm.conv_filter.sum() == 1.0 and cm.conv_filter[2, 2, 0, 0] == 1.0 and cm.has_initial_filter()
    cm.set_trainable(False)
    assert cm.trainable is False
    assert Zero()(np.zeros((3, 7))).shape == (3, 1)

    class Stub(ConvLayer):           # the classification needs no device: skip the prior factorisation
        def _build_prior_cholesky(self):
            pass
    rng = np.random.default_rng(0)
    view = FullView((9, 9), 5, 3, 2)
    mk = lambda mf: Stub(RBF(view.patch_length, 1.0, 1.0), mf, PatchInducingFeatures(rng.standard_normal((6, view.patch_length))),   # noqa: E731
                         view, gp_count=4, q_mu=np.zeros((6, 4)), q_sqrt=np.tile(np.eye(6), (4, 1, 1)))
    assert mk(cm).identity_mean and mk(cm).generic_mean is None
    assert mk("conv2d").identity_mean and not mk(None).identity_mean and mk(Zero()).generic_mean is None
    moved = Conv2dMean(5, 3, 4, stride=2)
    moved.conv_filter[0, 0, 1, 2] = 0.5
    assert not mk(moved).identity_mean and mk(moved).generic_mean is moved            # a changed filter goes through __call__
    assert not mk(Conv2dMean(5, 3, 4, stride=1)).identity_mean                        # another geometry than the view's
    assert mk(idm).generic_mean is idm
    with pytest.raises(ValueError):
        mk("identity").generic_mean


def test_kernel_host_classes_validate_without_a_device():
    """Constructor-level behaviour of the gpflow stand-ins that needs no GPU: parameter validation, ARD expansion,
    the {type, variance, p1, p2} description pushed to the device model, InducingPoints length."""
    from deepcgp_amd.kernels import RBF, ArcCosine, InducingPoints
    k = RBF(4, variance=2.0, lengthscales=3.0)
    assert k._describe() == [0.0, 2.0, 3.0, 0.0] and not k.ARD
    ka = RBF(3, variance=1.5, lengthscales=5.0, ARD=True)            # scalar initial value -> one lengthscale per dimension
    assert ka.lengthscales.shape == (3,) and np.all(ka.lengthscales == 5.0) and ka._describe() == [0.0, 1.5, 1.0, 0.0]
    assert np.allclose(RBF(2, lengthscales=[1.0, 4.0], ARD=True)._scaled(np.array([[2.0, 2.0]])), [[2.0, 0.5]])
    with pytest.raises(ValueError):
        RBF(2, variance=-1.0)
    with pytest.raises(ValueError):
        RBF(2, lengthscales=[1.0, 0.0], ARD=True)
    with pytest.raises(ValueError):
        ka._scaled(np.zeros((2, 5)))                                  # wrong input length
    a = ArcCosine(7, order=0)                                         # gpflow defaults (conv_gp/models.py:119)
    assert a._describe() == [1.0, 1.0, 1.0, 1.0] and np.all(a.Kdiag(np.zeros((3, 7))) == 1.0)
    with pytest.raises(NotImplementedError):
        ArcCosine(7, order=1)
    with pytest.raises(ValueError):
        ArcCosine(7, weight_variances=0.0)
    assert len(InducingPoints(np.zeros((5, 2)))) == 5


def test_synthetic_spec_variants():
    """make_spec's kernel variants: acos conv layers carry base = 'acos', the dense head carries per-dimension lengthscales."""
    from deepcgp_amd import synthetic as syn
    s1 = syn.make_spec((12, 12, 1), [(3, 1, 2)], (3, 1), M=6, S=2, base_kernel="acos")
    assert s1["convs"][0]["base"] == "acos" and s1["head"].get("kernel", "conv") == "conv"
    s2 = syn.make_spec((12, 12, 1), [(3, 2, 2)], (3, 1), M=6, S=2, head_kernel="rbf")
    h = s2["head"]
    assert h["kernel"] == "rbf" and h["Z"].shape == (6, 5 * 5 * 2) and h["ls_ard"].shape == (50,) and h["w"].shape == (1,)
    assert syn.layer_output_dims(s2) == [5 * 5 * 2, 10]


def test_optimiser_schedules_match_the_reference_formulas():
    """learning_rate = tf.train.exponential_decay(lr, step, decay_steps, 0.1, staircase=True) and the NatGrad gamma
    schedule min((step / 100 * 1e-3 + gamma0) * 0.2 ** steps_back, 1) of conv_gp/experiment.py:71-81."""
    from deepcgp_amd.models import learning_rate, natgrad_gamma
    assert learning_rate(0.01, 0, 50000) == 0.01
    assert learning_rate(0.01, 49999, 50000) == 0.01
    assert abs(learning_rate(0.01, 50000, 50000) - 0.001) < 1e-18
    assert abs(learning_rate(0.01, 125000, 50000) - 0.0001) < 1e-18
    assert natgrad_gamma(0) == 0.001
    assert abs(natgrad_gamma(5000, 0.001) - (50 * 1e-3 + 0.001)) < 1e-15
    assert abs(natgrad_gamma(5000, 0.001, steps_back=2) - (50 * 1e-3 + 0.001) * 0.04) < 1e-15
    assert natgrad_gamma(10 ** 9) == 1.0


def test_shard_range_of_the_library_equals_the_host_arithmetic():
    """dcgp_shard_range (the cut of a layer's parameter block in exchange mode 1 of the multi-rank training step; pure host arithmetic, no
    device needed) == deepcgp_amd.dist.grad_shard_range; the shards tile the block, are equally long up to the last, and world * shard
    covers it with at most world - 1 values of padding (the gradient blocks carry 64)."""
    import ctypes as C
    from deepcgp_amd import device as dev
    from deepcgp_amd.dist import grad_shard_range
    L = dev.lib()
    for n in (0, 1, 2, 7, 8, 9, 1000, 1001, 65536 * 10 + 2819):
        for world in (1, 2, 3, 8, 64):
            prev = 0
            for r in range(world):
                lo, hi, sh = C.c_long(), C.c_long(), C.c_long()
                assert L.dcgp_shard_range(n, world, r, C.byref(lo), C.byref(hi), C.byref(sh)) == 0
                assert (lo.value, hi.value, sh.value) == grad_shard_range(n, world, r)
                assert lo.value == prev and hi.value - lo.value <= sh.value
                prev = hi.value
            assert prev == n and sh.value * world - n <= max(world - 1, 0)
    assert L.dcgp_shard_range(10, 4, 4, C.byref(lo), C.byref(hi), None) != 0       # rank out of range


def _dry(args, timeout=90, launcher=False):
    import json as _json
    import time as _time
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("DCGP_RDZV_FILE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", args[1], "--master-addr", "127.0.0.1",
               "--master-port", "29741", os.path.join(ROOT, "bench.py")] + args
    t0 = _time.time()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out.returncode, (_json.loads(lines[-1]) if lines else None), out.stderr, _time.time() - t0


def test_bench_dry_run_walks_the_multi_rank_set_up_and_its_failure_paths():
    """`bench.py --gpus N --dry-run` (no GPU here: device calls simulated, the control flow is the product's): rendezvous, RCCL id broadcast,
    dcgp_comm_init_rank, the collective vote, one all-reduce on the agreed path -- and with injected faults the run either agrees on the
    host fallback or ends with a non-zero status in bounded time; nothing hangs."""
    rc, line, err, _ = _dry(["--gpus", "3", "--dry-run"])
    assert rc == 0 and line["dry_run"] and line["comm"] == "rccl" and line["n_gpus"] == 3 and len(line["events"]) == 5, (rc, line, err)
    for fault in ("no_rccl:1", "no_rccl:0", "init_fail:2"):                      # one rank without RCCL: every rank falls back together
        rc, line, err, _ = _dry(["--gpus", "3", "--dry-run", "--inject", fault])
        assert rc == 0 and line["comm"] == "host" and line["ranks_seen_by_rccl"] == 0, (fault, rc, line, err)
        assert "voting for the host all-reduce" in err
    rc, line, err, dt = _dry(["--gpus", "2", "--dry-run", "--inject", "late_id:1.5"])      # the id arrives late: the others wait for it
    assert rc == 0 and line["comm"] == "rccl" and line["events"][1]["t_s"] >= 1.5, (rc, line, err)
    rc, line, err, dt = _dry(["--gpus", "3", "--dry-run", "--inject", "die_before_init:1"])   # a rank dies before the rendezvous
    assert rc == 7 and line is None and dt < 40, (rc, line, err, dt)
    rc, line, err, _ = _dry(["--gpus", "2", "--dry-run", "--inject", "init_fail:1"], launcher=True, timeout=180)   # the driver's launcher
    assert rc == 0 and line["comm"] == "host", (rc, line, err)
