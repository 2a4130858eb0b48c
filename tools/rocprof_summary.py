#!/usr/bin/env python3
"""Summarises a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as a stats table:
   python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                      "max(vgpr_count), max(lds_size), max((grid_x*1.0/workgroup_x)*(grid_y*1.0/workgroup_y)*(grid_z*1.0/workgroup_z)), max(workgroup_x) "
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# total kernel time %.1f us over %d dispatches" % (tot / 1e3, sum(r[1] for r in rows)))
    print("%-78s %7s %11s %9s %9s %9s %6s %5s %7s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                          "pct", "vgpr", "lds_B", "wgs"))
    for r in rows:
        nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
        nm = re.sub(r"void ", "", nm)
        nm = re.sub(r"\(.*", "", nm)[:78]
        print("%-78s %7d %11.1f %9.2f %9.2f %9.2f %6.1f %5d %7d %6d" % (nm, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                        r[5] / 1e3, 100.0 * r[2] / tot, r[6] or 0,
                                                                        r[7] or 0, int(r[8] or 0)))


if __name__ == "__main__":
    main(sys.argv[1])
