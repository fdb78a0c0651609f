"""HBM traffic and L2 hit rate per launch of the hash variant's kernels from three rocprofv3 PMC passes.

    python tools/pmc_hash.py <fetch.csv> <write.csv> <tcc.csv> > profiles/pmc_hash.json

    rocprofv3 --kernel-trace --pmc FETCH_SIZE              --output-format csv -- python tools/gpu_check.py time_hash
    rocprofv3 --kernel-trace --pmc WRITE_SIZE              --output-format csv -- python tools/gpu_check.py time_hash
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -- python tools/gpu_check.py time_hash
(separate passes: the TCC block has four slots, FETCH_SIZE takes three).  gfx950 correction (MI355X_MICROARCH.md, HBM):
FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled (an UPPER bound for the gather kernels, whose reads
are not wide); WRITE_SIZE as is.  Both in KiB.  bench.py puts these into the hash rooflines' `traffic` / `l2_hit_rate`."""
import csv
import json
import sys
from collections import defaultdict


def avg(path, counter):
    acc = defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return {k: sum(v[2:]) / len(v[2:]) if len(v) > 4 else sum(v) / len(v) for k, v in acc.items()}


import os  # noqa: E402
KERNELS = tuple(os.environ.get("NGM_PMC_KERNELS", "k_render_fwd,k_hash_,k_grad_reduce,k_field_bwd,k_stash_bwd").split(","))
fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
hit, miss = avg(sys.argv[3], "TCC_HIT_sum"), avg(sys.argv[3], "TCC_MISS_sum")
out = {}
for k in sorted(set(fetch) | set(hit)):
    if not any(s in k for s in KERNELS):
        continue
    f, w, h, m = fetch.get(k), write.get(k), hit.get(k), miss.get(k)
    out[k] = dict(FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, hbm_bytes=None if f is None else int((2 * f + (w or 0.0)) * 1024),
                  TCC_HIT_sum=h, TCC_MISS_sum=m, l2_hit_rate=None if not h and not m else h / (h + m))
json.dump(dict(command=os.environ.get("NGM_PMC_COMMAND", "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} (three passes) "
                       "--output-format csv -- python tools/gpu_check.py time_hash   [M1 batch, hash 16 x 2 + 1 x 32 network]"),
               correction="hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE halves wide coalesced reads, "
                          "MI355X_MICROARCH.md; for gather kernels the factor 2 is an upper bound)", kernels=out), sys.stdout, indent=1)
print()
