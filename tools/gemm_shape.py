#!/usr/bin/env python
"""usage (GPU box): python tools/gemm_shape.py M N K [batch] [layout nn|tn|nt|tt]  -- one product of csrc/gemm_gen.hip under each output tile
(ctx option gemm_tile): which configuration a shape of the reverse pass should take"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import device as dev          # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
mode = sys.argv[5] if len(sys.argv) > 5 else "nn"
ctx = dev.get_context()
L = dev.lib()
rng = np.random.default_rng(0)
dA = ctx.to_device(rng.standard_normal((batch, M * K)))
dB = ctx.to_device(rng.standard_normal((batch, K * N)))
dC = ctx.to_device(np.zeros((batch, M * N)))
a_rs, a_cs = (1, M) if mode[0] == "t" else (K, 1)
b_rs, b_cs = (1, K) if mode[1] == "t" else (N, 1)
for tile in (0, 32, 64, 128):
    with ctx.options(gemm_tile=tile):
        def go():
            ctx._check(L.dcgp_gemm_strided(ctx.handle, dA.ptr, a_rs, a_cs, M * K, dB.ptr, b_rs, b_cs, K * N, dC.ptr, N, M * N, M, N, K, batch, 1.0, 0,
                                           None, 0, 0, None, 0, 0, 0))
        for _ in range(3):
            go()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            go()
        us = 1e6 * (time.perf_counter() - t0) / n
    print("%d x %d x %d batch %d %s  tile %3d: %8.1f us (call incl. sync)  %6.1f TF/s" % (M, N, K, batch, mode, tile, us, 2.0 * M * N * K * batch / us / 1e6))
