#!/usr/bin/env python
"""bench.py -- forward ELBO steps/sec of the conv-GP hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--scaling weak|strong]

One "step" = one forward ELBO evaluation of a synthetic minibatch (compute_log_likelihood semantics:
S = 10 samples, all layers, data term + all KLs, scalar read back to the host), inputs already resident
in HBM, nothing cached across steps.  At N = 1 the workload is BASELINE.json configs[1] (MNIST 1-layer
M = 256, batch 32) in its conv-layer + head form (K_uf + Cholesky + conditional); noise comes from the
counter-based device RNG, as the reference draws it inside its graph.

N > 1 (launched by torch.distributed.run, one rank per GPU): the minibatch images are sharded over the
ranks; every rank runs the same step on its shard and ONE RCCL all-reduce (1 x fp64) of the data term per
step joins them.  --scaling weak (default): 32 images per GPU (global batch 32*N), value counts
batch-32-equivalent steps (images/s / 32); --scaling strong: global batch fixed at 32.
Rank 0 prints ONE JSON line.  torch is used only as host plumbing (gloo barrier / broadcast / max).
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

from deepcgp_amd import device as dev                    # noqa: E402
from deepcgp_amd import synthetic as syn                 # noqa: E402
from deepcgp_amd.dist import shard_range, init_rccl      # noqa: E402
from deepcgp_amd.models import build_from_spec           # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix peak (datasheet; the guide lists no fp64 row): 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


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
    """The oracle (NumPy/OpenBLAS fp64 restatement in the reference's operation order) timed on the host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_build import oracle_model
    spec, X, Y = syn.make_config(name, S=S)
    X, Y = X[:batch], Y[:batch]
    model = oracle_model(spec, X, Y)
    rng = np.random.default_rng(0)
    small = oracle_model(spec, X[:2], Y[:2])
    small.num_samples = 1
    small.compute_log_likelihood(X[:2], Y[:2], rng=rng)          # warm BLAS / page in
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and (not times or time.perf_counter() - t_all + times[-1] < budget_s):
        t0 = time.perf_counter()
        model.compute_log_likelihood(X, Y, rng=rng)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "ELBO steps/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d full forward ELBO steps of the same workload (batch %d, S=%d) with the float64 NumPy/OpenBLAS "
                      "oracle in the reference's operation order; median %.2f s/step; not TensorFlow" % (len(times), batch, S, med)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=str, default="cfg2_mnist_CH_M256", choices=sorted(syn.CONFIGS))
    ap.add_argument("--scaling", type=str, default="weak", choices=["weak", "strong"])
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--dedup-layer0", action="store_true",
                    help="evaluate layer 0 on the distinct images only (exact; off by default so that the step does "
                         "the same work as the reference, which tiles the batch S times)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad-leg", action="store_true", help="skip the informational value-and-gradient timing")
    ap.add_argument("--profile", action="store_true",
                    help="for rocprofv3 runs: exactly --warmup + --steps steps, none of the extra regions, no CPU baseline")
    ap.add_argument("--comm", type=str, default="rccl", choices=["rccl", "gloo"],
                    help="N > 1: rccl = in-stream ncclAllReduce inside dcgp_elbo_forward (default); gloo = host "
                         "all-reduce of the per-rank data term (debug / fallback when RCCL cannot initialise)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    td = None
    if world > 1:
        import torch
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group("gloo", rank=rank, world_size=world)

    ctx = dev.Context(local_rank if dev.device_count() > local_rank else 0)
    dev._default_ctx = ctx
    if world > 1:
        def bcast(b):
            t = torch.tensor(list(b), dtype=torch.uint8)
            td.broadcast(t, src=0)
            return bytes(t.tolist())
        comm = args.comm
        if comm == "rccl":
            ok = 1
            try:
                init_rccl(ctx, rank, world, bcast)
            except Exception as exc:          # noqa: BLE001 -- any failure must reach the collective vote below
                ok = 0
                print("rank %d: RCCL init failed (%s); falling back to the host all-reduce" % (rank, exc), file=sys.stderr)
            vote = torch.tensor([ok], dtype=torch.int32)
            td.all_reduce(vote, op=td.ReduceOp.MIN)
            if int(vote[0]) == 0:
                dev.lib().dcgp_comm_destroy(ctx.handle)
                comm = "gloo(fallback)"
    else:
        comm = "none"

    cfg = syn.CONFIGS[args.config]
    S = args.samples
    per_rank_batch = cfg["batch"]
    if args.scaling == "weak":
        global_batch = cfg["batch"] * world
        lo, hi = rank * cfg["batch"], (rank + 1) * cfg["batch"]
    else:
        global_batch = cfg["batch"]
        lo, hi = shard_range(global_batch, rank, world)
        per_rank_batch = hi - lo
    seed = 1234 + list(syn.CONFIGS).index(args.config)
    spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=S, num_data=cfg["num_data"], seed=seed)
    Xg, Yg = syn.make_batch(cfg["hwc"], global_batch, seed=seed)
    model = build_from_spec(spec, Xg[lo:hi], Yg[lo:hi])
    model.dedup_layer0 = bool(args.dedup_layer0)
    dX, dY = ctx.to_device(Xg[lo:hi]), ctx.to_device(Yg[lo:hi], np.int32)
    scale = float(spec["num_data"]) / float(global_batch)

    def step(i):
        if comm.startswith("gloo"):
            # host-side join: the local data term comes back, is summed over ranks with gloo, ELBO assembled here
            _, data, kl = model.compute_log_likelihood(dX, dY, seed=i, scale=scale, return_parts=True)
            t = torch.tensor([data], dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.SUM)
            return float(t[0]) * scale - kl
        return model.compute_log_likelihood(dX, dY, seed=i, scale=scale)

    def run_steps(first, n, depth):
        """n steps; depth > 1 keeps that many steps queued (dcgp_elbo_forward_enqueue / _collect): every step's ELBO
        still comes back to the host, the host just does not wait for step i before queueing step i + 1."""
        if depth <= 1 or comm.startswith("gloo"):
            v = None
            for i in range(n):
                v = step(first + i)
            return v
        tickets, v = [], None
        for i in range(n):
            tickets.append(model.enqueue_log_likelihood(dX, dY, seed=first + i, scale=scale))
            if len(tickets) >= depth:
                v = model.collect_log_likelihood(tickets.pop(0))
        while tickets:
            v = model.collect_log_likelihood(tickets.pop(0))
        return v

    def barrier():
        ctx.sync()
        if td is not None:
            td.barrier()
        ctx.sync()

    # HIP events bracket only the two roofline kernels (gemm_cond_s3, kuf) on their launch stream; the mode is
    # switched on before the warm-up so that every lazy first-use cost of the event path is paid outside the timed region
    ctx.timing_enable(2)
    elbo = None
    # The HIP runtime has a one-off ~50 ms hiccup somewhere in the first few dozen steps of a process (seen in 1 run
    # out of 4 when only a handful of warm-up steps were run); warm-up is untimed, so run at least 50 of them.
    for i in range(args.warmup if args.profile else max(args.warmup, 50)):
        elbo = step(i)
    ctx.timing_reset()
    gc.collect()
    gc.disable()       # a generation-2 collection inside the timed loop showed up as a ~50 ms hiccup in 1 run out of 4
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        elbo = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    timing = ctx.timing()
    ctx.timing_enable(0)
    # the same K steps with two steps in flight (throughput mode, same values) ...
    dt_pipe, pipe_matches = None, None
    if not args.profile and world == 1:   # informational leg, single GPU only (like the training-step legs)
        try:
            ctx.timing_enable(2)
            run_steps(0, 4, 2)
            barrier()
            tp = time.perf_counter()
            elbo_pipe = run_steps(args.warmup, args.steps, 2)
            barrier()
            dt_pipe = time.perf_counter() - tp
            pipe_matches = bool(elbo_pipe == elbo)   # same seeds: bit-identical to the synchronous loop
        except Exception as e:                        # never let an informational leg cost the bench line
            print("two-in-flight leg skipped: %r" % (e,), file=sys.stderr)
            dt_pipe = None
        finally:
            ctx.timing_enable(0)
    # the same K steps without any event bracket (what the instrumentation costs) ...
    barrier()
    t1 = time.perf_counter()
    for i in range(0 if args.profile else args.steps):
        step(args.warmup + i)
    barrier()
    dt_plain = time.perf_counter() - t1
    # ... with layer-0 de-duplication (propagate() tiles the batch S times, so layer 0 sees S identical copies;
    # evaluating the distinct images once is exact -- bit-identical ELBO) as an additional, separately labelled number
    dt_dedup = None
    if cfg["convs"] and not args.dedup_layer0 and not args.profile:
        model.dedup_layer0 = True
        for i in range(2):
            step(i)
        barrier()
        t2 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        barrier()
        dt_dedup = time.perf_counter() - t2
        model.dedup_layer0 = False
    # ... and once more with every kernel family bracketed, for the informational per-kernel table only
    ctx.timing_enable(1)
    ctx.timing_reset()
    for i in range(0 if args.profile else min(args.steps, 10)):
        step(args.warmup + i)
    barrier()
    timing_all = ctx.timing()
    ctx.timing_enable(0)
    # ... and the same step followed by its reverse pass (dcgp_elbo_grad: value and gradient with respect to every
    # trainable parameter -- what the reference's training step differentiates, experiment.py:84-108).  Informational,
    # single-rank only (the gradient all-reduce is not wired up yet); never part of `value`.
    dt_grad, dt_train, dt_train_dedup = None, None, None
    if world == 1 and not args.profile and not args.no_grad_leg:
        n_g = max(3, min(args.steps, 20))
        for i in range(2):
            model.compute_gradients(dX, dY, seed=i, scale=scale, fetch=False)
        barrier()
        t3 = time.perf_counter()
        for i in range(n_g):
            model.compute_gradients(dX, dY, seed=args.warmup + i, scale=scale, fetch=False)
        barrier()
        dt_grad = (time.perf_counter() - t3) / n_g
        # ... and a full optimisation step: value + gradient + the device Adam update (tf.train.AdamOptimizer semantics,
        # lr small enough to leave the benchmark state essentially where it was)
        barrier()
        t4 = time.perf_counter()
        for i in range(n_g):
            model.compute_gradients(dX, dY, seed=args.warmup + i, scale=scale, fetch=False)
            model.adam_step(1e-9, i + 1)
        barrier()
        dt_train = (time.perf_counter() - t4) / n_g
        # ... and the same optimisation step with the exact layer-0 de-duplication (forward and reverse pass of the first
        # layer on the N distinct images; identical gradients, tests/test_gpu_model.py)
        dt_train_dedup = None
        if cfg["convs"] and not args.dedup_layer0:
            model.dedup_layer0 = True
            for i in range(2):
                model.compute_gradients(dX, dY, seed=i, scale=scale, fetch=False)
            barrier()
            t5 = time.perf_counter()
            for i in range(n_g):
                model.compute_gradients(dX, dY, seed=args.warmup + i, scale=scale, fetch=False)
                model.adam_step(1e-9, n_g + i + 1)
            barrier()
            dt_train_dedup = (time.perf_counter() - t5) / n_g
            model.dedup_layer0 = False
    if td is not None:
        t = torch.tensor([dt, dt_plain, dt_dedup or 0.0], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt, dt_plain = float(t[0]), float(t[1])
        dt_dedup = float(t[2]) if dt_dedup is not None else None

    if rank == 0:
        units_per_step = global_batch / float(cfg["batch"])        # batch-32-equivalent ELBO steps per step
        value = units_per_step * args.steps / dt
        out = {
            # BASELINE.json's metric string for its configs[1]; `value` is the steps/sec part (forward ELBO evaluations,
            # batch-32-equivalent when sharded), the K_uf HBM GB/s part is `kuf_hbm_gbs` / `roofline_kuf`
            "metric": "ELBO steps/sec (batch=%d) + achieved HBM GB/s on K_uf, MNIST M=256 1-layer" % cfg["batch"]
                      if args.config.startswith("cfg2") else "ELBO steps/sec (batch=%d) + achieved HBM GB/s on K_uf" % cfg["batch"],
            "value": value, "unit": "ELBO steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.config, "variant": "conv layer + head" if cfg["convs"] else "head only",
                       "M": cfg["M"], "per_gpu_batch": per_rank_batch, "global_batch": global_batch, "num_samples": S,
                       "image": list(cfg["hwc"]), "layers": len(cfg["convs"]) + 1, "noise": "device Philox RNG",
                       "dedup_layer0": bool(args.dedup_layer0), "parallelism": "image-sharded x%d, %s all-reduce of 1 f64" % (world, comm)},
            "elbo": elbo,
            "ms_per_step_without_event_timing": 1e3 * dt_plain / args.steps,
            "ms_per_step_two_in_flight": None if dt_pipe is None else 1e3 * dt_pipe / args.steps,
            "steps_per_s_two_in_flight": None if dt_pipe is None else args.steps / dt_pipe,
            "two_in_flight_elbo_identical": pipe_matches,
            "steps_per_s_with_exact_layer0_dedup": (units_per_step * args.steps / dt_dedup) if dt_dedup else None,
            "value_and_grad_steps_per_s": (1.0 / dt_grad) if dt_grad else None,
            "value_and_grad_ms": (1e3 * dt_grad) if dt_grad else None,
            "train_step_ms_value_grad_adam": (1e3 * dt_train) if dt_train else None,
            "train_step_ms_with_exact_layer0_dedup": (1e3 * dt_train_dedup) if dt_train_dedup else None,
        }
        # ---- roofline of the dominant kernel: the R-batched L_q^T A product with fused square-reduce ----
        rows0 = per_rank_batch if (args.dedup_layer0 and cfg["convs"]) else per_rank_batch * S
        kern = {k: {"launches": v[0], "avg_us": 1e3 * v[1] / max(v[0], 1)} for k, v in timing_all.items()}
        out["kernel_times_us"] = {k: round(v["avg_us"], 2) for k, v in sorted(kern.items())}
        if cfg["convs"]:
            c = spec["convs"][0]
            P, L, Kc = conv_geometry(c, rows0)
            M, R = c["M"], c["R"]
            nl = len(cfg["convs"]) + 1
            # gemm_cond_s3 is launched once per layer; layer 0 dominates -- use its algorithmic flops only when it
            # is the only conv layer, else report the sum over layers per step
            flops_s3 = 0.0
            rows = rows0
            for ci, cc in enumerate(spec["convs"]):
                Pc, Lc, Kcc = conv_geometry(cc, rows if ci == 0 else per_rank_batch * S)
                flops_s3 += float(cc["R"]) * cc["M"] ** 2 * Kcc
            n_conv = len(cfg["convs"])
            t_s3 = timing.get("gemm_cond_s3", (0, 0.0))
            per_step_ms = t_s3[1] / max(args.steps, 1)
            ach = flops_s3 / (per_step_ms * 1e-3) / 1e12 if per_step_ms > 0 else None
            out["roofline"] = {"kernel": "gemm_tn_kernel<128,128,4,4> (stage 3: T_r = G_r^T A1, fused sum of squares; %d conv-layer launch(es)/step)" % n_conv,
                               "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": (ach / FP64_MFMA_PEAK_TFLOPS) if ach else None,
                               "traffic": pmc_traffic("gemm_cond_s3", args.config),
                               "measured_mfma_f64_ceiling_tflops": 76.5,
                               "algorithmic_flops_per_step": flops_s3, "ms_per_step_in_kernel": per_step_ms,
                               "note": "algorithmic flops = sum_layers R*M^2*K (triangular product counted as M^2 per column, SURVEY 8(d)); fp64 MFMA peak"}
            # K_uf sweep (layer 0): algorithmic bytes 8*(N'*H*W*C + M*L + P*M*N')
            bytes_kuf = 8.0 * (rows0 * c["H"] * c["W"] * c["C"] + M * L + float(P) * M * rows0)
            t_kuf = timing.get("kuf", (0, 0.0))
            if t_kuf[0] and n_conv == 1:
                us = 1e3 * t_kuf[1] / t_kuf[0]
                gbs = bytes_kuf / (us * 1e-6) / 1e9
                out["kuf_hbm_gbs"] = gbs
                out["roofline_kuf"] = {"kernel": "patch_rbf_kernel (K_uf sweep, layer 0; held to 2 workgroups/CU while it overlaps the factorisation chain)", "bound": "hbm", "achieved": gbs,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                       "traffic": pmc_traffic("kuf", args.config),
                                       "algorithmic_bytes_per_launch": bytes_kuf, "avg_us": us}
        if not args.no_cpu_baseline and not args.profile and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, S, cfg["batch"])
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if td is not None:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
