#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (gpurun_out/prof/*.db) into a small text table for profiles/.
usage: python tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name "
        "order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}", file=out)
    print(f"# {'kernel':100s} calls  total_ms    avg_ms    min_ms    max_ms   pct  vgpr agpr sgpr    lds   grid_x wg_x", file=out)
    for r in rows:
        print(f"{r[0][:102]:102s} {r[1]:5d} {r[2] / 1e6:9.3f} {r[3] / 1e6:9.3f} {r[4] / 1e6:9.3f} {r[5] / 1e6:9.3f} "
              f"{100 * r[2] / tot:5.1f} {r[6]:5d} {r[7]:4d} {r[8]:4d} {r[9]:6d} {r[10]:8d} {r[11]:4d}", file=out)


if __name__ == "__main__":
    main()
