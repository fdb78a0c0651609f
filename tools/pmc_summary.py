"""Average rocprofv3 PMC counters per kernel from counter_collection CSVs.
    python tools/pmc_summary.py gpurun_out/pmc_a/a_counter_collection.csv [more.csv ...]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if not any(s in k for s in ("k_field_bwd", "k_render_fwd", "k_stash_bwd", "k_grad_reduce", "k_composite", "k_field_points", "k_hash", "k_knn")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        v = v[2:] if len(v) > 4 else v
        print(f"   {c:28s} avg {sum(v) / len(v):16.1f}  (n={len(v)})")
