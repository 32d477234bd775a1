#!/usr/bin/env python
"""bench.py -- forward ELBO steps/sec of the conv-GP hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]

One "step" = one forward ELBO evaluation of a synthetic minibatch (compute_log_likelihood semantics:
S = 10 samples, all layers, data term + all KLs, scalar read back to the host), inputs already resident
in HBM, nothing cached across steps.  At N = 1 the workload is BASELINE.json configs[1] (MNIST 1-layer
M = 256, batch 32) in its conv-layer + head form (K_uf + Cholesky + conditional); noise comes from the
counter-based device RNG, as the reference draws it inside its graph.

N > 1: one process per GPU.  Launched by a process launcher (RANK / WORLD_SIZE in the environment, e.g.
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) each process is one rank; launched
plainly (`python bench.py --gpus N`) this process starts the N ranks itself.  The ranks find each other over a
small TCP host group (deepcgp_amd.dist.HostGroup -- no torch anywhere), rank 0's RCCL id is handed round, and
from then on the only exchange per step is ONE in-stream ncclAllReduce (1 x fp64) of the data term.  The
minibatch images are sharded over the ranks:
  * `value` is STRONG scaling -- the global batch stays the configuration's (32 at cfg2), which is what
    north_star's ">= 6x at 8 GPUs over 1 GPU" and SURVEY 8(e) define;
  * `weak` (config batch per GPU, value = batch-equivalent steps/s) is reported beside it in the same line.
Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deepcgp_amd import device as dev                                   # noqa: E402
from deepcgp_amd import synthetic as syn                                # noqa: E402
from deepcgp_amd.dist import HostGroup, shard_range, init_rccl, spawn_ranks   # noqa: E402
from deepcgp_amd.models import build_from_spec                          # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix peak (datasheet; the guide lists no fp64 row): 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
PMC_SOURCE = ("profiles/pmc_traffic.json: HBM-side bytes per launch from the COMMITTED rocprofv3 --pmc passes of tools/collect_profiles.sh "
              "(FETCH_SIZE x 2 + WRITE_SIZE), not re-measured by this run")


def pmc_traffic(kernel, config):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 [gfx950 correction] +
    WRITE_SIZE, KB -> bytes; profiles/pmc_traffic.json), or None when no pass exists for this kernel/config."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            return json.load(fh).get(config, {}).get(kernel)
    except (OSError, ValueError):
        return None


def conv_geometry(c, rows):
    P = ((c["H"] - c["f"]) // c["s"] + 1) * ((c["W"] - c["f"]) // c["s"] + 1)
    return P, c["f"] * c["f"] * c["C"], rows * P


def cpu_baseline(name, S, batch, budget_s=30.0):
    """The oracle (NumPy/OpenBLAS fp64 restatement in the reference's operation order) timed on the host, as BASELINE.md section 2
    lays it out: 2 warm-up + >= 5 timed forward ELBO steps of the same workload, median; the BLAS thread count is the best of
    {8, 32, 64, all cores} (one probing step each -- 256 OpenBLAS threads on a 0.2 s GEMM are slower than 32) and is reported; a
    1-thread row beside it on a quarter of the batch (the step is linear in the batch: value scaled by 1/4 and labelled).  Bounded
    at ~budget_s seconds of CPU work whatever the host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_build import oracle_model
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                            # no control over the pool: time it as it comes
        threadpool_limits = None
    spec, X, Y = syn.make_config(name, S=S)
    X, Y = X[:batch], Y[:batch]
    model = oracle_model(spec, X, Y)
    rng = np.random.default_rng(0)
    small = oracle_model(spec, X[:2], Y[:2])
    small.num_samples = 1
    small.compute_log_likelihood(X[:2], Y[:2], rng=rng)          # warm BLAS / page in

    def one(m, Xb, Yb):
        t0 = time.perf_counter()
        m.compute_log_likelihood(Xb, Yb, rng=rng)
        return time.perf_counter() - t0

    def limited(n):
        import contextlib
        return threadpool_limits(limits=n, user_api="blas") if threadpool_limits else contextlib.nullcontext()

    one(model, X, Y)                                             # warm-up 1: first touch of every buffer (several times a steady step)
    t_begin = time.perf_counter()
    ncpu = os.cpu_count() or 1
    probes = {}
    for n in sorted({min(c, ncpu) for c in (8, 32, 64, ncpu)}):    # one probing step per thread count: warm-ups 2...
        with limited(n):
            probes[n] = one(model, X, Y)
        if time.perf_counter() - t_begin > 0.4 * budget_s:
            break
    best = min(probes, key=probes.get)
    times = []
    with limited(best):
        one(model, X, Y)
        while len(times) < 5 or (len(times) < 9 and time.perf_counter() - t_begin < 0.5 * budget_s):
            times.append(one(model, X, Y))
            if time.perf_counter() - t_begin > 0.8 * budget_s and len(times) >= 5:
                break
    med = float(np.median(times))
    q = max(batch // 4, 1)
    m1 = oracle_model(spec, X[:q], Y[:q])
    with limited(1):
        t1 = [one(m1, X[:q], Y[:q]) for _ in range(2)][-1] if threadpool_limits else None
    out = {"value": 1.0 / med, "unit": "ELBO steps/s", "cores": best, "kind": "port",
           "threads": best, "host_cores": ncpu, "steps_timed": len(times), "warmup_steps": len(probes) + 2,
           "thread_probe_s_per_step": {str(k): round(v, 3) for k, v in probes.items()},
           "sample": "%d full forward ELBO steps of the same workload (batch %d, S=%d) with the float64 NumPy/OpenBLAS oracle in the "
                     "reference's operation order (materialised K_uf, per-patch triangular solves, dense tensordot), %d BLAS threads "
                     "(best of %s on %d host cores), after %d warm-up steps; median %.2f s/step; not TensorFlow"
                     % (len(times), batch, S, best, sorted(probes), ncpu, len(probes) + 2, med)}
    # the same arithmetic batched the way a CPU likes it (oracle/fast_cpu.py: ONE triangular solve over all K columns, ONE (R M) x M x K
    # GEMM instead of the reference's loop of 2 P solves + tensordot): what a tuned CPU implementation would reach, beside the
    # reference-order row above
    try:
        from oracle import fast_cpu
        fb = {}
        for n in sorted(probes):
            with limited(n):
                fast_cpu.elbo(spec, X, Y, rng=rng)
                t0 = time.perf_counter(); fast_cpu.elbo(spec, X, Y, rng=rng); fb[n] = time.perf_counter() - t0
            if time.perf_counter() - t_begin > 1.3 * budget_s:
                break
        nb = min(fb, key=fb.get)
        with limited(nb):
            ft = []
            while len(ft) < 5 and (len(ft) < 3 or time.perf_counter() - t_begin < 1.6 * budget_s):
                t0 = time.perf_counter(); fast_cpu.elbo(spec, X, Y, rng=rng); ft.append(time.perf_counter() - t0)
        out["best_cpu"] = {"value": 1.0 / float(np.median(ft)), "unit": "ELBO steps/s", "cores": nb, "kind": "port",
                           "thread_probe_s_per_step": {str(k): round(v, 3) for k, v in fb.items()},
                           "sample": "%d steps of the same workload, oracle/fast_cpu.py (batched: one trsm over all K columns, one (R M) x M x K GEMM; "
                                     "equal to the reference-order oracle to 1e-10, tests/test_oracle_cpu.py), %d BLAS threads; median %.3f s/step"
                                     % (len(ft), nb, float(np.median(ft)))}
    except Exception as exc:                                       # a baseline row must never break the bench line
        out["best_cpu"] = {"error": repr(exc)}
    if t1:
        out["value_1thread"] = 1.0 / (t1 * batch / q)
        out["sample_1thread"] = ("1 BLAS thread, second of 2 steps on a quarter of the batch (%d images): %.2f s, scaled by %d (the step is "
                                 "linear in the batch)" % (q, t1, batch // q))
    return out


def dry_run(args, rank, world, local_rank):
    """`bench.py --gpus N --dry-run [--inject ...]`: the multi-rank set-up of main() and nothing else, with faults on request, so that the
    first real 8-GPU run cannot hang silently.  Every wait is bounded (DCGP_HOSTGROUP_TIMEOUT seconds, default 120; the dry run uses 20);
    a rank that cannot go on says why on stderr and exits non-zero -- spawn_ranks / the launcher then stops the others.  Rank 0 prints one
    JSON line: which communication path the ranks agreed on and what each step of the walk took."""
    faults = {}
    for f in filter(None, args.inject.split(",")):
        k, _, v = f.partition(":")
        faults.setdefault(k, []).append(float(v) if k == "late_id" else int(v))
    t0 = time.perf_counter()
    events = []

    def mark(what):
        events.append({"step": what, "t_s": round(time.perf_counter() - t0, 3)})
    if rank in faults.get("die_before_init", []):
        print("rank %d: injected fault: dying before the rendezvous" % rank, file=sys.stderr)
        return 7
    os.environ.setdefault("DCGP_HOSTGROUP_TIMEOUT", "20")
    try:
        grp = HostGroup(rank, world)
    except (TimeoutError, OSError) as exc:
        print("rank %d: rendezvous failed: %r" % (rank, exc), file=sys.stderr)
        return 3
    mark("host group of %d ranks met" % world)
    try:
        n_dev = dev.device_count()
    except Exception:                                   # noqa: BLE001 -- no library / no GPU: the walk is simulated
        n_dev = 0
    ctx = None
    if n_dev > 0:
        ctx = dev.Context(local_rank if n_dev > local_rank else local_rank % n_dev)
    comm, rccl_ranks, ok = "host", 0, 1
    try:
        if rank in faults.get("no_rccl", []):
            raise RuntimeError("injected fault: librccl could not be loaded")
        if rank == 0:
            for d in faults.get("late_id", []):
                time.sleep(d)
            uid = dev.comm_unique_id() if ctx is not None else bytes(range(128))
        else:
            uid = bytes(128)
        uid = grp.broadcast_bytes(uid)
        mark("RCCL id broadcast (%d bytes)" % len(uid))
        if rank in faults.get("init_fail", []):
            raise RuntimeError("injected fault: ncclCommInitRank failed")
        if ctx is not None:
            ctx.comm_init(world, rank, uid)         # (two ranks on one device: RCCL refuses -- the real failure path)
        mark("dcgp_comm_init_rank")
    except Exception as exc:                            # noqa: BLE001 -- any failure must reach the collective vote below
        ok = 0
        print("rank %d: RCCL init failed (%s); voting for the host all-reduce" % (rank, exc), file=sys.stderr)
        if rank in faults.get("no_rccl", []):
            grp.broadcast_bytes(b"")                    # (keep the group's exchanges in step: the others are in the id broadcast)
    try:
        agreed = int(grp.allreduce([ok], "min")[0])
        mark("vote: %s" % ("RCCL on every rank" if agreed else "some rank without RCCL -> host all-reduce on all"))
        if agreed:
            comm = "rccl"
            rccl_ranks = ctx.comm_count() if ctx is not None else world
        elif ctx is not None:
            ctx.comm_destroy()
        # one data-term-sized exchange on the agreed path (the host path is the group itself; the device path needs a GPU)
        total = float(grp.allreduce([float(rank + 1)], "sum")[0])
        assert total == world * (world + 1) / 2
        if comm == "rccl" and ctx is not None:
            buf = ctx.to_device(np.array([float(rank + 1)]))
            ctx.allreduce_sum(buf)
            assert float(buf.numpy()[0]) == total
        mark("one all-reduce on the agreed path")
        grp.barrier()
        grp.close()
    except Exception as exc:                            # noqa: BLE001
        print("rank %d: dry run failed behind the vote: %r" % (rank, exc), file=sys.stderr)
        return 4
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "comm": comm, "ranks_seen_by_rccl": rccl_ranks, "devices_visible": n_dev,
                          "simulated_device_calls": ctx is None, "faults": args.inject, "events": events}))
    return 0


class Leg:
    """One sharded workload on this rank: the model, its device-resident shard and the step function."""

    def __init__(self, ctx, grp, comm, cfg_name, S, global_batch, lo, hi, dedup):
        cfg = syn.CONFIGS[cfg_name]
        self.ctx, self.grp, self.comm = ctx, grp, comm
        self.global_batch, self.local_batch = global_batch, hi - lo
        seed = 1234 + list(syn.CONFIGS).index(cfg_name)
        self.spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=S, num_data=cfg["num_data"], seed=seed)
        Xg, Yg = syn.make_batch(cfg["hwc"], global_batch, seed=seed)
        self.Xh, self.Yh = Xg[lo:hi], Yg[lo:hi]
        self.model = build_from_spec(self.spec, self.Xh, self.Yh)
        self.model.dedup_layer0 = bool(dedup)
        self.model.global_batch = global_batch
        if hi - lo != global_batch:
            self.model.set_shard(lo, global_batch)   # noise drawn at the element's place in the un-sharded batch: same ELBO for any rank count
        self.dX, self.dY = ctx.to_device(self.Xh), ctx.to_device(self.Yh, np.int32)
        self.scale = float(self.spec["num_data"]) / float(global_batch)

    def step(self, i):
        if self.comm == "host":
            # fallback join (RCCL could not initialise, e.g. two ranks on one device): the local data term comes back,
            # is summed over the ranks by the host group, and the ELBO is assembled here
            _, data, kl = self.model.compute_log_likelihood(self.dX, self.dY, seed=i, scale=self.scale, return_parts=True)
            return float(self.grp.allreduce([data], "sum")[0]) * self.scale - kl
        return self.model.compute_log_likelihood(self.dX, self.dY, seed=i, scale=self.scale)

    def run_pipelined(self, first, n, depth):
        """n steps with `depth` steps queued (dcgp_elbo_forward_enqueue / _collect): every step's ELBO still comes back
        to the host, the host just does not wait for step i before queueing step i + 1."""
        tickets, v = [], None
        for i in range(n):
            tickets.append(self.model.enqueue_log_likelihood(self.dX, self.dY, seed=first + i, scale=self.scale))
            if len(tickets) >= depth:
                v = self.model.collect_log_likelihood(tickets.pop(0))
        while tickets:
            v = self.model.collect_log_likelihood(tickets.pop(0))
        return v

    def barrier(self):
        self.ctx.sync()
        self.grp.barrier()
        self.ctx.sync()

    def timed(self, first, n, fn=None):
        """Wall time of n steps between two (device sync + rank barrier) brackets, max over the ranks."""
        fn = fn or self.step
        self.barrier()
        t0 = time.perf_counter()
        v = None
        for i in range(n):
            v = fn(first + i)
        self.barrier()
        dt = time.perf_counter() - t0
        return float(self.grp.allreduce([dt], "max")[0]), v


def shard_sweep(leg, steps, warmup, depths=(1, 2)):
    """Strong-scaling preview on ONE GPU: the step on the shard a rank would hold at G = 1, 2, 4, 8 (same parameters, same
    num_data / global-batch scale), synchronous (`depth` 1) and with two steps in flight (`depth` 2: the parameter-only chain of
    step i + 1 runs under the data path of step i, so the replicated prefix leaves the critical path).  t(b) = replicated + b *
    per_image fitted through the end points gives the Amdahl fraction of the step that does not shrink with the shard
    (factorisation chain, KL, launch + host latency).  The all-reduce of one double is not in these numbers."""
    res = {}
    B = leg.local_batch
    for depth in depths:
        out = {}
        for G in (1, 2, 4, 8):
            b = max(1, -(-B // G))
            dX, dY = leg.ctx.to_device(leg.Xh[:b]), leg.ctx.to_device(leg.Yh[:b], np.int32)

            def run(first, n):
                tickets = []
                for i in range(n):
                    if depth == 1:
                        leg.model.compute_log_likelihood(dX, dY, seed=first + i, scale=leg.scale)
                    else:
                        tickets.append(leg.model.enqueue_log_likelihood(dX, dY, seed=first + i, scale=leg.scale))
                        if len(tickets) >= depth:
                            leg.model.collect_log_likelihood(tickets.pop(0))
                while tickets:
                    leg.model.collect_log_likelihood(tickets.pop(0))
            run(0, 60 if steps >= 60 else max(steps, 10))   # the uploads above left the device idle: past the clock ramp first
            leg.ctx.sync()
            t0 = time.perf_counter()
            run(warmup, steps)
            leg.ctx.sync()
            out[G] = (b, 1e3 * (time.perf_counter() - t0) / steps)
        (b1, t1), (b8, t8) = out[1], out[8]
        per_image = (t1 - t8) / max(b1 - b8, 1)
        replicated = t1 - per_image * b1
        res["synchronous" if depth == 1 else "two_in_flight"] = {
            "per_rank_batch_ms": {str(G): {"images": b, "ms_per_step": round(t, 4)} for G, (b, t) in out.items()},
            "replicated_ms": round(replicated, 4), "replicated_fraction": round(replicated / t1, 4),
            "predicted_strong_speedup_excluding_allreduce": {str(G): round(t1 / t, 3) for G, (b, t) in out.items()}}
    return res


def shard_sweep_all(ctx, grp, S, skip, steps=30):
    """The same preview for every other BASELINE configuration (BASELINE.json assigns cfg3 / cfg4 / cfg5 to 8 GPUs)."""
    out = {}
    for name in syn.CONFIGS:
        if name == skip:
            continue
        cfg = syn.CONFIGS[name]
        try:
            lg = Leg(ctx, grp, "none", name, S, cfg["batch"], 0, cfg["batch"], False)
            n = steps if cfg["M"] < 1024 or not cfg["convs"] else max(steps // 3, 8)
            out[name] = shard_sweep(lg, n, 5)
            lg.model.close()
        except Exception as e:                # noqa: BLE001 -- informational
            out[name] = {"skipped": repr(e)}
    return out


def kuf_measure(leg, ctx, steps):
    """K_uf of layer 0, measured two ways on the bench workload (device-resident inputs, the model's own step):
      * materialised: the sweep + GEMM route (ctx option no_fused_layer) really writes K_uf [M, N' P] to HBM and reads
        it back; its sweep launch ("kuf" timer, HIP events on the launch stream, every launch) gives achieved HBM GB/s on the
        algorithmic bytes -- this is `kuf_hbm_gbs`;
      * one-launch route (what `value` runs): K_uf never leaves the chip.  The shader-clock stamps of sampled workgroups give the time
        a strip spends in its K_uf phase; rounds x that is the launch's K_uf time, and bytes / time the materialised-EQUIVALENT rate --
        reported beside the true HBM bytes of that phase (images and Z in, nothing out)."""
    from deepcgp_amd import device as dev
    res = {}
    with ctx.options(no_fused_layer=1):
        # the sweep is enqueued in front of the factorisation chain and runs beside it (two workgroups per CU, the chain's waves at a
        # higher priority); option no_early_sweep puts it behind the chain: the launch alone on the chip, which is what the rate is quoted for
        n = max(20, min(steps, 60))
        for key, late in (("sweep_us", 1), ("sweep_us_beside_the_chain", 0)):
            with ctx.options(no_early_sweep=late):
                for i in range(30):
                    leg.step(i)
                ctx.timing_enable(3)
                ctx.timing_reset()
                for i in range(n):
                    leg.step(100 + i)
                leg.barrier()
                t = ctx.timing().get("kuf", (0, 0.0))
                ctx.timing_enable(0)
                if t[0]:
                    res[key] = 1e3 * t[1] / t[0]
                    res["sweep_launches_sampled"] = t[0]
        # every row evaluated (option kuf_no_rep): what the sweep costs when the rows do NOT repeat images -- layer 0 of this model sees
        # the batch tiled S times (DGP_Base.propagate), and by default a unit evaluates a tile once and stores it to every row showing
        # that image; the bytes written are the same
        with ctx.options(no_early_sweep=1, kuf_no_rep=1):
            for i in range(10):
                leg.step(i)
            ctx.timing_enable(3)
            ctx.timing_reset()
            for i in range(n):
                leg.step(100 + i)
            leg.barrier()
            t = ctx.timing().get("kuf", (0, 0.0))
            ctx.timing_enable(0)
            if t[0]:
                res["sweep_us_every_row_evaluated"] = 1e3 * t[1] / t[0]
    for i in range(3):
        leg.step(i)
    # phase stamps of the one-launch layer kernel (csrc/conv_fused.hip CF_TR): [8 sampled workgroups][16 waves][16 stamps]
    buf = ctx.to_device(np.zeros((8, 16, 16), np.int64), np.int64)
    dev.lib().dcgp_debug_set_fused_trace(ctx.handle, buf.ptr)
    leg.step(7)
    ctx.sync()
    dev.lib().dcgp_debug_set_fused_trace(ctx.handle, None)
    t = buf.numpy().astype(np.float64)
    phase, ghz = [], []
    for b in range(8):
        live = t[b, :, 0] > 0
        if not live.any() or not t[b, 0, 11] > t[b, 0, 10]:
            continue
        clk = (t[b, 0, 9] - t[b, 0, 0]) / ((t[b, 0, 11] - t[b, 0, 10]) / 100.0) / 1e3        # shader GHz from wall_clock64 (100 MHz)
        ghz.append(clk)
        # K_uf phase of the workgroup: from the last wave past the image / norm stage (stamp 1) to the last wave done with K_uf (stamp 2)
        phase.append((t[b, live, 2].max() - t[b, live, 1].max()) / clk / 1e3)
    if phase:
        res["fused_phase_us_per_strip"] = float(np.mean(phase))
        res["shader_ghz"] = float(np.mean(ghz))
    return res


def head_sweep_flops(h, rows):
    """ConvKernel.Kdiag N' P^2 (2L+4) [full count; the kernel evaluates tiles on and right of the diagonal] + ConvKernel.Kzx N' P M (2L+4)
    (conv_gp/kernels.py:106-133)."""
    P, L, _ = conv_geometry(h, 1)
    return float(rows) * P * P * (2 * L + 4) + float(rows) * P * h["M"] * (2 * L + 4)


HEAD_KERNEL = ("head_units_kernel (ConvKernel.Kzx: weighted patch sum reduced in-kernel, + ConvKernel.Kdiag: all patch pairs of an image; "
               "wave-sized units of one launch, norms folded into the MFMA, 17-instruction 2^t epilogue)")
HEAD_NOTE = ("N'*P^2*(2L+4) + N'*P*M*(2L+4) (SURVEY 8(d)); the Kdiag part evaluates the tiles on and right of the diagonal only.  On gfx950 the "
             "fp64 MFMA and every VALU instruction of a SIMD issue one after the other (tools/pipe_mix.hip), so the exp of each kernel value "
             "(17 VALU instructions = 68 cycles against 112 MFMA cycles per value at L = 25) is part of the bound: ~0.87 by this count is the ceiling")


def head_only_leg(ctx, grp, S, steps):
    """The reference's literal "1-layer M=256" (results/N60000_M256/options.toml:3: scalar M = SVGP head with the ConvKernel,
    no ConvLayer): forward ELBO steps/s and the roofline of its largest term, ConvKernel.Kdiag (kernels.py:106-115)."""
    leg = Leg(ctx, grp, "none", "cfg2_mnist_H_M256", S, 32, 0, 32, False)
    for i in range(200):     # building the model left the device idle: past the clock ramp before anything is timed (see main)
        leg.step(i)
    dt, _ = leg.timed(20, steps)
    ctx.timing_enable(1)
    ctx.timing_reset()
    for i in range(50):
        leg.step(i)
    ctx.sync()
    tim = ctx.timing()
    # the sweep alone on the chip (the step above runs it beside the factorisation chain, two workgroups per CU)
    with ctx.options(head_no_overlap=1):
        for i in range(3):
            leg.step(i)
        ctx.timing_reset()
        for i in range(50):
            leg.step(i)
        ctx.sync()
        tim_alone = ctx.timing()
    ctx.timing_enable(0)
    h = leg.spec["head"]
    rows = 32 * S
    flops = head_sweep_flops(h, rows)
    us_in_step = 1e3 * tim["head_sweep"][1] / max(tim["head_sweep"][0], 1)
    us = 1e3 * tim_alone["head_sweep"][1] / max(tim_alone["head_sweep"][0], 1)
    ach = flops / (us_in_step * 1e-6) / 1e12       # the launch as the step runs it (beside the chain): what `frac` reports
    ach_alone = flops / (us * 1e-6) / 1e12
    out = {"head_only_steps_per_s": steps / dt, "head_only_ms_per_step": 1e3 * dt / steps,
           "roofline_head": {"kernel": HEAD_KERNEL, "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic("head_sweep", "cfg2_mnist_H_M256"), "traffic_source": PMC_SOURCE,
                             "algorithmic_flops_per_launch": flops, "avg_us": us_in_step, "launches_sampled": tim["head_sweep"][0],
                             "avg_us_alone_on_the_chip": us, "achieved_alone_on_the_chip": ach_alone, "frac_alone_on_the_chip": ach_alone / FP64_MFMA_PEAK_TFLOPS,
                             "note": HEAD_NOTE + ".  avg_us / achieved / frac: the launch inside the head-only step (head_only_steps_per_s), where it runs "
                                     "beside the factorisation chain on a side stream, two workgroups per CU, and the step is the longer of the two; "
                                     "*_alone_on_the_chip: the same launch with the chain first and the sweep behind it (ctx option head_no_overlap)"},
           "head_only_kernel_times_us": {k: round(1e3 * v[1] / max(v[0], 1), 2) for k, v in sorted(tim.items())}}
    leg.model.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=str, default="cfg2_mnist_CH_M256", choices=sorted(syn.CONFIGS))
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--dedup-layer0", action="store_true",
                    help="evaluate layer 0 on the distinct images only (exact; off by default so that the step does "
                         "the same work as the reference, which tiles the batch S times)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad-leg", action="store_true", help="skip the informational value-and-gradient timing")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the head-only and shard-sweep legs")
    ap.add_argument("--no-all-configs", action="store_true", help="skip the strong-scaling preview of the other BASELINE configurations")
    ap.add_argument("--profile", action="store_true",
                    help="for rocprofv3 runs: exactly --warmup + --steps steps, none of the extra regions, no CPU baseline")
    ap.add_argument("--dry-run", action="store_true",
                    help="N > 1 readiness check: walk the multi-rank set-up only -- rendezvous of the host group, broadcast of the RCCL id, "
                         "dcgp_comm_init_rank on every rank, the collective vote and the host-join fallback -- print one JSON line and exit; "
                         "no model, no timing.  Without a GPU the device calls are simulated (the control flow is what is walked)")
    ap.add_argument("--inject", type=str, default="",
                    help="--dry-run only, comma-separated faults: die_before_init:R (rank R exits before the rendezvous), late_id:SECONDS "
                         "(rank 0 draws the id late), no_rccl:R (librccl missing on rank R), init_fail:R (ncclCommInitRank fails on rank R)")
    ap.add_argument("--comm", type=str, default="rccl", choices=["rccl", "host"],
                    help="N > 1: rccl = in-stream ncclAllReduce inside dcgp_elbo_forward (default); host = host-group "
                         "all-reduce of the per-rank data term (debug / fallback when RCCL cannot initialise)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process is the launcher -- one child per rank, rank 0 prints the line
        raise SystemExit(spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    if args.dry_run:
        raise SystemExit(dry_run(args, rank, world, local_rank))
    grp = HostGroup(rank, world)
    n_dev = dev.device_count()
    ctx = dev.Context(local_rank if n_dev > local_rank else local_rank % max(n_dev, 1))
    dev._default_ctx = ctx
    comm, rccl_ranks = "none", 0
    if world > 1:
        comm = args.comm
        if comm == "rccl":
            ok = 1
            try:
                init_rccl(ctx, rank, world, grp.broadcast_bytes)
            except Exception as exc:          # noqa: BLE001 -- any failure must reach the collective vote below
                ok = 0
                print("rank %d: RCCL init failed (%s); falling back to the host all-reduce" % (rank, exc), file=sys.stderr)
            if int(grp.allreduce([ok], "min")[0]) == 0:
                ctx.comm_destroy()
                comm = "host"
            else:
                rccl_ranks = ctx.comm_count()

    cfg = syn.CONFIGS[args.config]
    S = args.samples
    # strong scaling (value): the configuration's batch, sharded; weak: the configuration's batch on every rank
    lo, hi = shard_range(cfg["batch"], rank, world)
    if int(grp.allreduce([hi - lo], "min")[0]) == 0:
        raise SystemExit("--gpus %d leaves a rank without an image at batch %d" % (world, cfg["batch"]))
    leg = Leg(ctx, grp, comm, args.config, S, cfg["batch"], lo, hi, args.dedup_layer0)
    model, spec = leg.model, leg.spec

    # The timed loop carries no instrumentation at all.  Warm-up is untimed, so it is made long enough for two things measured on this part
    # (tools/bench_overhead.py): a one-off 13-50 ms hiccup of the HIP runtime somewhere in the first ~70 steps of a process, and the
    # clock ramp after ANY idle stretch of a few milliseconds -- the 20-40 steps behind one run 3-7 % slow (0.90 -> 0.86 -> 0.836 ms).  The
    # garbage collection (tens of ms of idle device) therefore comes BEFORE the warm-up, and the timed region follows the warm-up with
    # nothing but the sync + rank barrier between them.
    gc.collect()
    gc.disable()       # a generation-2 collection inside the timed loop showed up as a ~50 ms hiccup in 1 run out of 4
    for i in range(args.warmup if args.profile else max(args.warmup, 100)):
        leg.step(i)
    dt, elbo = leg.timed(args.warmup, args.steps)
    # The roofline kernel's launch duration: HIP events on its own stream around EVERY launch of it (timing mode 3), in a loop of
    # its own (>= 50 steps) behind the timed region -- the judged value pays nothing for it, and the average does not depend on --steps.
    n_roof = 0 if args.profile else max(50, min(args.steps, 200))
    ctx.timing_enable(3)
    ctx.timing_reset()
    for i in range(n_roof):
        leg.step(args.warmup + i)
    leg.barrier()
    timing = ctx.timing()
    ctx.timing_enable(0)
    extra = {}

    def informational(name, fn):
        """An informational leg can never cost the bench line."""
        if args.profile:
            return None
        try:
            return fn()
        except Exception as e:                # noqa: BLE001
            print("%s leg skipped: %r" % (name, e), file=sys.stderr)
            return None

    # the same K steps with two steps in flight (throughput mode, same values)
    def pipelined():
        leg.run_pipelined(0, 4, 2)
        d, v = leg.timed(args.warmup, 1, lambda first: leg.run_pipelined(first, args.steps, 2))
        return d, bool(v == elbo)   # same seeds: bit-identical to the synchronous loop
    pipe = informational("two-in-flight", pipelined) if comm != "host" else None
    dt_plain = dt      # (kept for readers of earlier rounds' lines: the timed loop IS the un-instrumented one now)

    # weak scaling beside it: the configuration's batch on every rank
    weak = None
    if world > 1 and not args.profile:
        def weak_leg():
            wl = Leg(ctx, grp, comm, args.config, S, cfg["batch"] * world, rank * cfg["batch"], (rank + 1) * cfg["batch"], args.dedup_layer0)
            for i in range(20):
                wl.step(i)
            d, _ = wl.timed(args.warmup, args.steps)
            wl.model.close()
            return {"value": world * args.steps / d, "unit": "batch-%d-equivalent ELBO steps/s" % cfg["batch"], "ms_per_step": 1e3 * d / args.steps,
                    "global_batch": cfg["batch"] * world, "per_gpu_batch": cfg["batch"]}
        weak = informational("weak-scaling", weak_leg)

    # exact layer-0 de-duplication (propagate() tiles the batch S times, so layer 0 sees S identical copies; evaluating the
    # distinct images once is exact -- bit-identical ELBO): an additional, separately labelled number
    dt_dedup = None
    if cfg["convs"] and not args.dedup_layer0:
        def dedup():
            model.dedup_layer0 = True
            try:
                for i in range(2):
                    leg.step(i)
                return leg.timed(args.warmup, args.steps)[0]
            finally:
                model.dedup_layer0 = False
        dt_dedup = informational("dedup", dedup)

    # Evaluation sweeps at ONE parameter state (the reference's LogLikelihoodLogger / AccuracyLogger, conv_gp/utils/log.py:55-68, run hundreds of batches
    # between two optimiser steps): with dcgp_model_set_factor_reuse(2) a step whose parameters were not written since the last factorisation is its
    # data path only.  A separately labelled number -- `value` above never runs in this mode (the reference's step recomputes the chain).
    eval_fixed = None
    if world == 1:
        def eval_leg():
            model.set_factor_reuse(2)
            try:
                for i in range(5):
                    leg.step(i)
                skips0 = model.chain_skips
                d, v = leg.timed(args.warmup, args.steps)
                return {"steps_per_s": args.steps / d, "ms_per_step": 1e3 * d / args.steps, "elbo_identical": bool(v == elbo),
                        "steps_that_skipped_the_chain": model.chain_skips - skips0, "steps": args.steps}
            finally:
                model.set_factor_reuse(1)
        eval_fixed = informational("eval-at-fixed-parameters", eval_leg)

    # once more with every kernel family bracketed, for the informational per-kernel table only
    timing_all = {}
    if not args.profile:
        ctx.timing_enable(1)
        ctx.timing_reset()
        for i in range(min(args.steps, 10)):
            leg.step(args.warmup + i)
        leg.barrier()
        timing_all = ctx.timing()
        ctx.timing_enable(0)

    # the same step followed by its reverse pass (dcgp_elbo_grad: value and gradient with respect to every trainable
    # parameter -- what the reference's training step differentiates, experiment.py:84-108).  Informational, single-rank.
    grad = {}
    if world == 1 and not args.no_grad_leg:
        def grad_leg():
            g = {}
            n_g = max(3, min(args.steps, 20))
            cg = lambda i: model.compute_gradients(leg.dX, leg.dY, seed=i, scale=leg.scale, fetch=False)   # noqa: E731
            for i in range(2):
                cg(i)
            g["value_and_grad_ms"] = 1e3 * leg.timed(args.warmup, n_g, cg)[0] / n_g

            def train_step(i):   # value, gradient and the Adam update: one device call (dcgp_model_train_step_adam)
                model.train_step(leg.dX, leg.dY, 1e-9, seed=i, scale=leg.scale)     # lr small enough to leave the benchmark state essentially where it was
            g["train_step_ms_value_grad_adam"] = 1e3 * leg.timed(args.warmup, n_g, train_step)[0] / n_g
            if cfg["convs"] and not args.dedup_layer0:
                model.dedup_layer0 = True
                try:
                    for i in range(2):
                        cg(i)
                    g["train_step_ms_with_exact_layer0_dedup"] = 1e3 * leg.timed(args.warmup, n_g, train_step)[0] / n_g
                finally:
                    model.dedup_layer0 = False
            # roofline of the reverse pass's two chip-filling product kernels (first conv layer, tiled batch): conv_bwd_fused
            # (dK_uf = inv(L)^T [sum_r (S_r A1) o 2 gv_r + ...]: R dense M x M x K products + one triangular) and the symmetric
            # W_r = 2 A1 diag(gv_r) A1^T contraction (R x the lower half of M x M x K), timed by HIP events on their own streams in a
            # loop of their own; they overlap in the step (different streams), so the sum of their times is an upper bound of their share
            if cfg["convs"]:
                # (ctx option grad_nofork: the reverse pass on ONE stream, so that each of the two kernels is timed alone on the chip -- in the
                # step proper they run beside each other on two streams and share it, which is what train_step_ms measures)
                with ctx.options(grad_nofork=1):
                    cg(999)
                    ctx.timing_enable(1)
                    ctx.timing_reset()
                    for i in range(min(n_g, 6)):
                        cg(1000 + i)
                    leg.barrier()
                    tim = ctx.timing()
                    ctx.timing_enable(0)
                # ... and the same two launches inside the step as it runs (two streams: they share the chip, each one's events bracket the other's work too)
                ctx.timing_enable(1)
                ctx.timing_reset()
                for i in range(min(n_g, 6)):
                    cg(2000 + i)
                leg.barrier()
                tim_step = ctx.timing()
                ctx.timing_enable(0)
                c0 = cfg["convs"][0]
                P0 = ((cfg["hwc"][0] - c0[0]) // c0[1] + 1) * ((cfg["hwc"][1] - c0[0]) // c0[1] + 1)
                Kc, M, R = leg.local_batch * S * P0, cfg["M"], c0[2]
                f_bwd = (2.0 * R + 1.0) * M * M * Kc          # dense S_r products 2 M^2 K each, the closing triangular product M^2 K
                f_wr = 1.0 * R * M * M * Kc                   # symmetric: M^2 K per output
                tb, tw = tim.get("conv_bwd_fused"), tim.get("grad_wr")
                if tb and tw and tb[0] and tw[0]:
                    us_b, us_w = 1e3 * tb[1] / tb[0], 1e3 * tw[1] / tw[0]
                    ach = (f_bwd + f_wr) / ((us_b + us_w) * 1e-6) / 1e12
                    in_step = None
                    sb, sw = tim_step.get("conv_bwd_fused"), tim_step.get("grad_wr")
                    if sb and sw and sb[0] and sw[0]:
                        ub, uw = 1e3 * sb[1] / sb[0], 1e3 * sw[1] / sw[0]
                        in_step = {"conv_bwd_fused_avg_us": ub, "w_r_contraction_avg_us": uw,
                                   "frac_if_fully_overlapped": (f_bwd + f_wr) / (max(ub, uw) * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                   "frac_if_one_after_the_other": (f_bwd + f_wr) / ((ub + uw) * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                   "note": "the two launches as the training step runs them, beside each other on two streams: each one's HIP events bracket "
                                           "part of the other's work, so the pair's share of the peak lies between the two figures"}
                    g["roofline_train"] = {"kernel": "conv_bwd_fused_kernel + syrk_kscale_kernel (W_r = 2 A1 diag(gv_r) A1^T, with its split-k reduction): the reverse pass's products of the first conv layer",
                                           "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                                           "traffic": None, "algorithmic_flops": f_bwd + f_wr,
                                           "conv_bwd_fused": {"avg_us": us_b, "flops": f_bwd, "tflops": f_bwd / us_b / 1e6},
                                           "w_r_contraction": {"avg_us": us_w, "flops": f_wr, "tflops": f_wr / us_w / 1e6},
                                           "launches_sampled": int(tb[0]),
                                           "in_step": in_step,
                                           "note": "flops: (2 R + 1) M^2 K + R M^2 K (symmetric / triangular products counted as M^2 per column, as SURVEY 8(d) "
                                                   "counts the forward's); times: HIP events around each launch with the reverse pass on one stream (ctx option "
                                                   "grad_nofork), i.e. each kernel alone on the chip; in the step they overlap on two streams"}
            return g
        grad = informational("gradient", grad_leg) or {}

    kuf_raw = None
    if cfg["convs"] and world == 1 and not args.profile:
        kuf_raw = informational("kuf", lambda: kuf_measure(leg, ctx, args.steps))

    def kuf_info(bytes_kuf, rows0, c, M, L, P, fused):
        o = {}
        if not kuf_raw:
            return o
        true_bytes = 8.0 * (rows0 * c["H"] * c["W"] * c["C"] + M * L)
        if "sweep_us" in kuf_raw:
            gbs = bytes_kuf / (kuf_raw["sweep_us"] * 1e-6) / 1e9
            o["kuf_hbm_gbs"] = gbs
            store_ceiling = informational("store-ceiling", lambda: ctx.measured_store_gbs(rows0, P, M))
            every = kuf_raw.get("sweep_us_every_row_evaluated")
            o["roofline_kuf"] = {"kernel": "head_units_kernel<.., storing form> (K_uf sweep of layer 0, materialised [M, N'P] in HBM: the sweep + GEMM route, "
                                           "ctx option no_fused_layer, timed in situ on the bench workload, the launch alone on the chip)",
                                 "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "traffic": pmc_traffic("kuf", args.config), "traffic_source": PMC_SOURCE,
                                 "algorithmic_bytes_per_launch": bytes_kuf,
                                 "avg_us": kuf_raw["sweep_us"], "avg_us_in_step_beside_the_chain": kuf_raw.get("sweep_us_beside_the_chain"),
                                 "launches_sampled": kuf_raw.get("sweep_launches_sampled"),
                                 "measured_store_ceiling_gbs": store_ceiling,
                                 "frac_of_measured_store_ceiling": (gbs / store_ceiling) if store_ceiling else None,
                                 "rows_sharing_an_image_evaluated_once": True,
                                 "every_row_evaluated": None if not every else {"avg_us": every, "achieved": bytes_kuf / (every * 1e-6) / 1e9,
                                                                                  "frac": bytes_kuf / (every * 1e-6) / 1e9 / HBM_PEAK_GBS},
                                 "note": "every byte of K_uf [M, S*N*P] is written each launch (true bytes = algorithmic bytes).  Layer 0 sees the batch "
                                         "tiled S times (DGP_Base.propagate), so S rows show each image: a unit evaluates a tile once and stores it to "
                                         "those S rows; `every_row_evaluated` is the same launch with that switched off (ctx option kuf_no_rep).  "
                                         "measured_store_ceiling_gbs: a pure store kernel writing the same matrix in the same tile pattern, timed in "
                                         "this run (csrc/peaks.hip)"}
        if fused and "fused_phase_us_per_strip" in kuf_raw:
            strips = -(-(rows0 * P) // 64)
            rounds = -(-strips // 256)
            us = kuf_raw["fused_phase_us_per_strip"] * rounds
            o["kuf_one_launch_route"] = {
                "note": "the route `value` runs never writes K_uf: a 64-column strip of it lives in LDS from the patch gather to the sample "
                        "(conv_fused_kernel phase 1).  materialised_equivalent_gbs = the reference's K_uf bytes / the time the launch spends in that "
                        "phase; true_hbm_bytes = what the phase really moves (images and Z in, nothing out)",
                "materialised_equivalent_gbs": bytes_kuf / (us * 1e-6) / 1e9, "materialised_equivalent_bytes": bytes_kuf,
                "true_hbm_bytes": true_bytes, "true_hbm_gbs": true_bytes / (us * 1e-6) / 1e9,
                "phase_us_per_strip": kuf_raw["fused_phase_us_per_strip"], "strips": strips, "rounds_of_256_cus": rounds, "phase_us_per_launch": us,
                "shader_ghz": kuf_raw.get("shader_ghz")}
        return o

    if world == 1 and not args.no_extra_legs:
        extra.update(informational("shard-sweep", lambda: {"strong_scaling_preview": shard_sweep(leg, min(args.steps, 100), args.warmup)}) or {})
        if not args.no_all_configs:
            extra.update(informational("shard-sweep-all", lambda: {"strong_scaling_preview_other_configs": shard_sweep_all(ctx, grp, S, args.config)}) or {})
        if args.config.startswith("cfg2"):
            extra.update(informational("head-only", lambda: head_only_leg(ctx, grp, S, min(args.steps, 100))) or {})

    # the fp64 MFMA rate this device sustains right now (csrc/peaks.hip: 4 waves per SIMD of back-to-back v_mfma_f64_16x16x4_f64, ~85 ms),
    # measured behind everything that is timed -- the ceiling the roofline fractions can also be read against
    mfma_ceiling = None if args.profile else informational("mfma-ceiling", ctx.measured_mfma_f64_tflops)

    if rank == 0:
        value = args.steps / dt
        per_rank_batch = hi - lo
        out = {
            # BASELINE.json's metric string for its configs[1]; `value` is the steps/sec part (forward ELBO evaluations of the
            # configuration's minibatch, sharded over the ranks), the K_uf HBM GB/s part is `kuf_hbm_gbs` / `roofline_kuf`
            "metric": "ELBO steps/sec (batch=%d) + achieved HBM GB/s on K_uf, MNIST M=256 1-layer" % cfg["batch"]
                      if args.config.startswith("cfg2") else "ELBO steps/sec (batch=%d) + achieved HBM GB/s on K_uf" % cfg["batch"],
            "value": value, "unit": "ELBO steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_steps_run": args.warmup if args.profile else max(args.warmup, 100),   # untimed: at least 100 (clock ramp, see the comment at the warm-up loop)
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.config, "variant": "conv layer + head" if cfg["convs"] else "head only",
                       "M": cfg["M"], "per_gpu_batch": per_rank_batch, "global_batch": cfg["batch"], "num_samples": S,
                       "image": list(cfg["hwc"]), "layers": len(cfg["convs"]) + 1, "noise": "device Philox RNG",
                       "dedup_layer0": bool(args.dedup_layer0),
                       "parallelism": "image-sharded x%d, %s" % (world, {"none": "no collective", "rccl": "in-stream ncclAllReduce of 1 f64",
                                                                          "host": "host-group all-reduce of 1 f64 (RCCL fallback)"}[comm])},
            "ranks_seen_by_rccl": rccl_ranks,
            "weak": weak if world > 1 else {"value": value, "unit": "batch-%d-equivalent ELBO steps/s" % cfg["batch"], "ms_per_step": 1e3 * dt / args.steps,
                                            "global_batch": cfg["batch"], "per_gpu_batch": cfg["batch"]},
            "elbo": elbo,
            "ms_per_step_without_event_timing": None if dt_plain is None else 1e3 * dt_plain / args.steps,
            "ms_per_step_two_in_flight": None if pipe is None else 1e3 * pipe[0] / args.steps,
            "steps_per_s_two_in_flight": None if pipe is None else args.steps / pipe[0],
            "two_in_flight_elbo_identical": None if pipe is None else pipe[1],
            "steps_per_s_with_exact_layer0_dedup": (args.steps / dt_dedup) if dt_dedup else None,
            "eval_steps_per_s_at_fixed_parameters": None if not eval_fixed else eval_fixed["steps_per_s"],
            "eval_at_fixed_parameters": eval_fixed and dict(eval_fixed, note="forward ELBO steps with the parameter-only chain (operand preparation, factorisations, "
                                                            "inverses, G / alpha, KL pieces) of the first step kept while no parameter is written "
                                                            "(dcgp_model_set_factor_reuse(2)): evaluation sweeps only, never `value`"),
            "value_and_grad_steps_per_s": (1e3 / grad["value_and_grad_ms"]) if grad.get("value_and_grad_ms") else None,
            "value_and_grad_ms": grad.get("value_and_grad_ms"),
            "train_step_ms_value_grad_adam": grad.get("train_step_ms_value_grad_adam"),
            "train_step_ms_with_exact_layer0_dedup": grad.get("train_step_ms_with_exact_layer0_dedup"),
            # the training step as a first-class number: value + gradient + Adam update in ONE device call (conv_gp/experiment.py:84-108)
            "train_steps_per_s": (1e3 / grad["train_step_ms_value_grad_adam"]) if grad.get("train_step_ms_value_grad_adam") else None,
            "train_steps_per_s_with_exact_layer0_dedup": (1e3 / grad["train_step_ms_with_exact_layer0_dedup"]) if grad.get("train_step_ms_with_exact_layer0_dedup") else None,
            "roofline_train": grad.get("roofline_train"),
        }
        out.update(extra)
        # ---- roofline of the dominant kernel -------------------------------------------------------------------------
        rows0 = per_rank_batch if (args.dedup_layer0 and cfg["convs"]) else per_rank_batch * S
        kern = {k: {"launches": v[0], "avg_us": 1e3 * v[1] / max(v[0], 1)} for k, v in timing_all.items()}
        out["kernel_times_us"] = {k: round(v["avg_us"], 2) for k, v in sorted(kern.items())}
        if cfg["convs"]:
            c = spec["convs"][0]
            P, L, Kc = conv_geometry(c, rows0)
            M = c["M"]
            # The conv layers' conditional.  One-launch route (conv_fused_kernel, M <= 256): K_uf, A1 = inv(L) K_uf, T_r = G_r^T A1,
            # mean and sample of a column strip in one workgroup; algorithmic flops per layer (a triangular M x M
            # product counts M^2 per column) = K M^2 (1 + R) + 2 M R K + K M (2 L + 4).  Sweep + GEMM route: the R-batched
            # second product (gemm_cond_s3) alone, R M^2 K.
            flops_s3 = flops_fused = 0.0
            for ci, cc in enumerate(spec["convs"]):
                Pc, Lc, Kcc = conv_geometry(cc, rows0 if ci == 0 else per_rank_batch * S)
                flops_s3 += float(cc["R"]) * cc["M"] ** 2 * Kcc
                flops_fused += float(Kcc) * cc["M"] ** 2 * (1 + cc["R"]) + 2.0 * cc["M"] * cc["R"] * Kcc + float(Kcc) * cc["M"] * (2 * Lc + 4)
            n_conv = len(cfg["convs"])
            fused = timing.get("conv_fused", (0, 0.0))[0] > 0
            t_dom = timing.get("conv_fused" if fused else "gemm_cond_s3", (0, 0.0))
            flops_dom = flops_fused if fused else flops_s3
            # (every launch of the kernel in the sampling loop is timed: average launch x launches per step)
            per_step_ms = (t_dom[1] / t_dom[0]) * (t_dom[0] / max(n_roof, 1)) if t_dom[0] and n_roof else 0.0
            ach = flops_dom / (per_step_ms * 1e-3) / 1e12 if per_step_ms > 0 else None
            out["roofline"] = {"kernel": ("conv_fused_kernel<4,2,2,1024> (whole conv layer of a 64-column strip per workgroup: patch sweep, inv(L) K_uf, "
                                          "R x G_r^T A1 with fused sums of squares, mean, sample; %d launch(es)/step)" % n_conv) if fused else
                                         ("gemm_tn_kernel<128,128,4,4> (stage 3: T_r = G_r^T A1, fused sum of squares; %d conv-layer launch(es)/step)" % n_conv),
                               "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": (ach / FP64_MFMA_PEAK_TFLOPS) if ach else None,
                               "traffic": pmc_traffic("conv_fused" if fused else "gemm_cond_s3", args.config), "traffic_source": PMC_SOURCE,
                               "measured_mfma_f64_ceiling_tflops": mfma_ceiling,
                               "frac_of_measured_ceiling": (ach / mfma_ceiling) if (ach and mfma_ceiling) else None,
                               "algorithmic_flops_per_step": flops_dom, "ms_per_step_in_kernel": per_step_ms,
                               "launches_sampled": t_dom[0], "sampling": "HIP events on the launch stream around every launch of the kernel, in a loop of %d steps behind the timed region" % n_roof,
                               "stage3_only_flops_per_step": flops_s3,
                               "traffic_note": ("the layer's own HBM bytes are the images in and the samples out (29.5 MB at cfg2: 31 MB measured until round 6's first "
                                                "session); since the prologues-ahead deal (DESIGN 4i) the counters also see 256 strips of A1 (132 KB each) written by the partial "
                                                "first round's spare workgroups and fetched back by the workgroups that run their second product: ~35 MB each way plus the "
                                                "write-through of the sc1 stores, Infinity-Cache resident, bought for 20 us of the launch -- not a re-read of the layer's inputs"
                                                ) if fused else None,
                               "note": "algorithmic flops: triangular products counted as M^2 per column (SURVEY 8(d)); peak = 78.6 TFLOP/s fp64 MFMA at the "
                                       "2.4 GHz datasheet clock.  (Earlier rounds read a shader clock of 2.1-2.3 GHz off wall_clock64 inside the kernel and called it "
                                       "power-limited; rocm-smi beside the looping step reads sclk 2398 MHz at 983 W of the 1400 W cap, and three strips of 446 k cycles "
                                       "accounted for the 572 us launch of that build at 2.39 GHz: the part is not power-bound here -- profiles/r06_power_and_clock.txt.  "
                                       "Since the prologues-ahead deal a CU runs whole + whole + fetched strips, 187 + 186 + 163 us: profiles/r06_fused_phase_trace.txt)"}
            # ---- the K_uf half of the metric (layer 0; SURVEY 8(d), conv_gp/layers.py:23-32) ----
            # algorithmic bytes of the sweep as the reference runs it: images in, Z in, K_uf [P, M, N'] out
            bytes_kuf = 8.0 * (rows0 * c["H"] * c["W"] * c["C"] + M * L + float(P) * M * rows0)
            if kuf_info:
                out.update(kuf_info(bytes_kuf, rows0, c, M, L, P, fused))
        if not cfg["convs"] and timing.get("head_sweep", (0, 0.0))[0]:
            t_h = timing["head_sweep"]
            flops_h = head_sweep_flops(spec["head"], per_rank_batch * S)
            us_h = 1e3 * t_h[1] / t_h[0]
            ach_h = flops_h / (us_h * 1e-6) / 1e12
            out["roofline"] = {"kernel": HEAD_KERNEL, "bound": "mfma", "achieved": ach_h, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach_h / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic("head_sweep", args.config), "traffic_source": PMC_SOURCE,
                               "algorithmic_flops_per_step": flops_h, "ms_per_step_in_kernel": us_h * 1e-3, "launches_sampled": t_h[0], "note": HEAD_NOTE}
        if not args.no_cpu_baseline and not args.profile and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, S, cfg["batch"])
        print("\n" + json.dumps(out))      # on a line of its own whatever a library (RCCL's banner) left on stdout before it
        sys.stdout.flush()
    grp.barrier()
    grp.close()


if __name__ == "__main__":
    main()
