"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel table.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--skip N]   (skip first N calls per kernel = warm-up)
"""
import sqlite3
import sys


def main(path, skip=0):
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x "
                      "from kernels order by start").fetchall()
    agg = {}
    for name, dur, vg, ag, sg, lds, gx, wx in rows:
        short = name.split("(")[0]
        a = agg.setdefault(short, dict(d=[], vg=vg, ag=ag, sg=sg, lds=lds, grid=gx, wg=wx))
        a["d"].append(dur)
    tot = sum(sum(a["d"][skip:]) for a in agg.values()) or 1
    print(f"{'kernel':70s} {'calls':>5s} {'avg_us':>9s} {'min_us':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'grid':>7s}")
    for k, a in sorted(agg.items(), key=lambda kv: -sum(kv[1]["d"][skip:])):
        d = a["d"][skip:] or a["d"]
        print(f"{k[:70]:70s} {len(d):5d} {sum(d) / len(d) / 1e3:9.2f} {min(d) / 1e3:9.2f} {100 * sum(d) / tot:6.1f} "
              f"{a['vg']:5d} {a['ag']:5d} {a['sg']:5d} {a['lds']:7d} {a['grid']:7d}")


if __name__ == "__main__":
    skip = 0
    if "--skip" in sys.argv:
        skip = int(sys.argv[sys.argv.index("--skip") + 1])
    main(sys.argv[1], skip)
