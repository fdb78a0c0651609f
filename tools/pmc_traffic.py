"""HBM traffic per launch of the render/backward kernels from two rocprofv3 PMC passes.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/pmc_field_bwd.json

FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC counter budget), so they come from separate runs of
    rocprofv3 --kernel-trace --pmc FETCH_SIZE   --output-format csv -- python tools/gpu_check.py time
    rocprofv3 --kernel-trace --pmc WRITE_SIZE   --output-format csv -- python tools/gpu_check.py time
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming
reads (16 B/lane, global_load and global_load_lds alike) -> doubled; WRITE_SIZE taken as is.  Both in KiB.
"""
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


fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if any(s in k for s in ("k_field_bwd", "k_render_fwd", "k_stash_bwd", "k_grad_reduce")):
        out[k] = dict(FETCH_SIZE_KB=fetch[k], WRITE_SIZE_KB=write.get(k, 0.0),
                      hbm_bytes=int((2 * fetch[k] + write.get(k, 0.0)) * 1024))
bwd = [k for k in out if "k_field_bwd" in k]
res = dict(kernel=bwd[0] if bwd else None,
           command="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) --output-format csv -- python tools/gpu_check.py time",
           correction="gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM); "
                      "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
           hbm_bytes_per_launch=out[bwd[0]]["hbm_bytes"] if bwd else None, all=out)
json.dump(res, sys.stdout, indent=1)
print()
