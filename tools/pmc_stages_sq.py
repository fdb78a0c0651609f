"""SQ counters per launch of the standalone stage kernels (three rocprofv3 --pmc passes over tools/stage_bench.py) and what
they say about the bound:  python tools/pmc_stages_sq.py <counter_collection.csv ...> > profiles/rNN_pmc_stages_sq.json
Derived per kernel (SQ_* cycle counters are summed over the SIMDs that ran waves; GRBM_GUI_ACTIVE = kernel duration in clocks):
  valu_per_wave       SQ_INSTS_VALU / SQ_WAVES
  valu_issue_frac     SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES     (share of busy SIMD cycles with a VALU instruction in flight)
  lds_issue_frac      SQ_ACTIVE_INST_LDS  / SQ_BUSY_CYCLES
  vmem_issue_frac     SQ_ACTIVE_INST_VMEM / SQ_BUSY_CYCLES
  wait_inst_frac      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        (share of wave-resident cycles spent waiting on a counter / dependency)
  wait_lds_frac       SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES"""
import csv
import json
import os
import sys
from collections import defaultdict

KERNELS = tuple(os.environ.get("NGM_PMC_KERNELS", "k_encode_points,k_composite_fwd,k_composite_bwd,k_sample_rays,k_fourier_wgrad,k_encode_bwd,k_hash_grad,k_field_points_fwd").split(","))
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(s in k for s in KERNELS):
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    c = {n: (sum(v[2:]) / len(v[2:]) if len(v) > 4 else sum(v) / len(v)) for n, v in cs.items()}
    g = lambda n: c.get(n)
    div = lambda a, b: (g(a) / g(b)) if g(a) is not None and g(b) else None
    out[k] = dict(counters=c, derived=dict(
        valu_per_wave=div("SQ_INSTS_VALU", "SQ_WAVES"), lds_per_wave=div("SQ_INSTS_LDS", "SQ_WAVES"), vmem_per_wave=div("SQ_INSTS_VMEM", "SQ_WAVES"),
        trans_per_wave=div("SQ_INSTS_VALU_TRANS", "SQ_WAVES"),
        valu_issue_frac=div("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"), lds_issue_frac=div("SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"),
        vmem_issue_frac=div("SQ_ACTIVE_INST_VMEM", "SQ_BUSY_CYCLES"), any_issue_frac=div("SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"),
        wait_inst_frac=div("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), wait_lds_frac=div("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"),
        lds_bank_conflict_frac=div("SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"), kernel_clocks=g("GRBM_GUI_ACTIVE")))
json.dump(dict(command="rocprofv3 --kernel-trace --pmc <six SQ counters> (three passes) --output-format csv -- python tools/stage_bench.py "
                       "[262144 rays x 128 samples; hash stages: 8 fields x 524288 points]", kernels=out), sys.stdout, indent=1)
print()
