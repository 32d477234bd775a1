#!/usr/bin/env python
"""Per-kernel counter averages out of the rocprofv3 --pmc passes of tools/collect_profiles.sh.
usage: python tools/pmc_summary.py <tag> [dir]   (reads <dir>/pmc_<tag>{a,b,c}/ and pmc_<tag>{f,w,s}_<config>/ ; dir = gpurun_out)"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
root = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
KEEP = ("conv_fused_kernel", "head_units_kernel", "head_sweep_kernel", "head_cond_kernel", "prep_solve_kernel", "elbo_tail_kernel", "chol_rl_kernel",
        "gemm_tn_kernel", "patch_rbf_kernel", "gemm_gen")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")


def collect(dirs):
    acc = defaultdict(lambda: defaultdict(list))   # (kernel, grid) -> counter -> values
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if not k.startswith(KEEP):
                    continue
                acc[(k, int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def dump(title, acc, top=None):
    print("== %s ==" % title)
    keys = sorted(acc, key=lambda kg: -max(sum(v) for v in acc[kg].values()))
    for kg in keys[:top]:
        print("  %s grid %d" % kg)
        for c in sorted(acc[kg]):
            v = acc[kg][c]
            print("      %-28s %.4g   (%d dispatches)" % (c, sum(v) / len(v), len(v)))


print("rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python bench.py --profile --steps 2 --warmup 1 [--config ...]   (1 x MI355X, ROCm 7.2;")
print("tools/collect_profiles.sh -> tools/pmc_bench.sh: DCGP_NO_SIDE_STREAM=1, counters in their own passes, kernel-trace only).  Values per dispatch (average).")
print("FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B: HBM-side read bytes = 2 x FETCH_SIZE KB (guide, HBM section).\n")
for cfg, note in (("cfg2_mnist_CH_M256", "headline: conv layer + head, M = 256, batch 32, S = 10"), ("cfg2_mnist_H_M256", "the reference's literal 1-layer: head only"),
                  ("cfg1_mnist_H_M32", "head only, M = 32"), ("cfg3_mnist_3layer_M256", "two conv layers + head"),
                  ("cfg4_cifar_3layer_M384", "sweep + GEMM route: M > 256"), ("cfg5_mnist_H_M1024", "head only, M = 1024, batch 128"),
                  ("cfg5_mnist_CH_M1024", "M = 1024, batch 128")):
    dirs = [os.path.join(root, "pmc_%s%s_%s" % (tag, s, cfg)) for s in "fws"]
    if not any(os.path.isdir(d) for d in dirs):
        continue
    print()
    dump("%s (%s)" % (cfg, note), collect(dirs), top=7)
