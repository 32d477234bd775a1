#!/usr/bin/env python
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh: HBM-side bytes per launch of the
kernels bench.py quotes a roofline for.  usage: python tools/pmc_traffic.py <tag> [dir]  -> JSON on stdout
(reads <dir>/pmc_<tag>f_<config>/ and pmc_<tag>w_<config>/; bytes = (2 x FETCH_SIZE [gfx950: 128-byte requests tallied at 64 B] +
WRITE_SIZE) x 1024, averaged over the dispatches of the kernel; a step with several launches of one kernel -- one per conv layer -- sums them)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
root = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"


def per_kernel(d, counter):
    acc = defaultdict(list)     # (short kernel name, grid) -> values
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]).replace("void ", "")
            acc[(k, int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    return acc


def bytes_of(fetch, write, pred):
    """sum over the (kernel, grid) groups selected by pred of the per-dispatch average traffic"""
    tot, hit = 0.0, False
    for key in set(fetch) | set(write):
        if not pred(*key):
            continue
        f = fetch.get(key, [0.0]); w = write.get(key, [0.0])
        tot += (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0
        hit = True
    return int(tot) if hit else None


out = {"_doc": "HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes (tools/collect_profiles.sh -> "
               "tools/pmc_bench.sh: one counter per pass, --kernel-trace only, bench.py --profile --steps 2 --warmup 1, DCGP_NO_SIDE_STREAM=1; "
               "built by tools/pmc_traffic.py).  FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte requests at 64 B).  "
               "A step with one launch of a kernel per conv layer sums them."}
for fd in sorted(d for d in glob.glob(os.path.join(root, "pmc_%sf_*" % tag)) if os.path.isdir(d)):
    cfg = os.path.basename(fd)[len("pmc_%sf_" % tag):]
    wd = os.path.join(root, "pmc_%sw_%s" % (tag, cfg))
    fetch, write = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    e = {}
    e["conv_fused"] = bytes_of(fetch, write, lambda k, g: k.startswith("conv_fused_kernel"))
    # head_units_kernel<NK4, TL, WMODE, NT>: WMODE 0 = the reducing form (head sweep), 1 / 2 = the storing form (K_uf sweep); NK4 > 0 =
    # the register-resident kernels of the short first-layer patches ("kuf"), NK4 == 0 = streamed operands, long patches ("kuf_long")
    def hu(k):
        m = re.match(r"head_units_kernel<\s*(\d+),\s*(\d+),\s*(\w+),", k)
        if not m:
            return None
        wm = {"false": 0, "true": 1}.get(m.group(3))
        return int(m.group(1)), int(m.group(3)) if wm is None else wm
    e["head_sweep"] = bytes_of(fetch, write, lambda k, g: hu(k) is not None and hu(k)[1] == 0)
    e["kuf"] = bytes_of(fetch, write, lambda k, g: (hu(k) is not None and hu(k)[1] != 0 and hu(k)[0] > 0) or k.startswith("patch_rbf_kernel"))
    e["kuf_long"] = bytes_of(fetch, write, lambda k, g: hu(k) is not None and hu(k)[1] != 0 and hu(k)[0] == 0)
    tn = sorted({g for (k, g) in (set(fetch) | set(write)) if k.startswith("gemm_tn_kernel<128")}, reverse=True)
    if tn:   # the R-batched second product is the largest grid of the 128-row tile kernel, the first product the next one
        e["gemm_cond_s3"] = bytes_of(fetch, write, lambda k, g: k.startswith("gemm_tn_kernel<128") and g == tn[0])
        if len(tn) > 1:
            e["gemm_cond_s1"] = bytes_of(fetch, write, lambda k, g: k.startswith("gemm_tn_kernel<128") and g == tn[1])
    out[cfg] = {k: v for k, v in e.items() if v is not None}
# passes taken with DCGP_NO_FUSED_LAYER=1 (<config>_unfused): the materialised K_uf sweep of a configuration whose step uses the one-launch layer
for cfg in [c for c in out if c.endswith("_unfused")]:
    e = out.pop(cfg)
    if "kuf" in e:
        out.setdefault(cfg[:-len("_unfused")], {})["kuf"] = e["kuf"]
print(json.dumps(out, indent=1))
