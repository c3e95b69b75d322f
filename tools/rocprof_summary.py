#!/usr/bin/env python3
"""Condense rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into the small text
summaries kept under profiles/.

  rocprof_summary.py stats <results.db>          per-kernel table (--kernel-trace --stats run)
  rocprof_summary.py pmc   <results.db> [...]     per-kernel counter means (--pmc runs), with the
                                                  copy-kernel calibration of FETCH_SIZE/WRITE_SIZE
"""
import sqlite3
import sys


def short(name, n=70):
    name = name.split("(")[0] if "xaac" in name else name
    return name if len(name) <= n else name[:n - 3] + "..."


def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                     "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-72s %6s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:12]:
        print("%-72s %6d %12.1f %10.2f %10.2f %10.2f %6.1f" % (short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3,
                                                              r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    for r in rows:
        if "xaac" in r[0]:
            print("\n%s: vgpr=%s agpr=%s sgpr=%s lds=%s B scratch=%s B grid=%s threads wg=%s" % (
                short(r[0]), r[6], r[7], r[8], r[9], r[10], r[11], r[12]))


def pmc(dbs):
    for db in dbs:
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from "
                         "counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
        print("# %s" % db)
        print("%-72s %-14s %6s %16s %10s" % ("kernel", "counter", "calls", "mean value", "avg_us"))
        for r in rows:
            if "xaac" in r[0] or "copy" in r[0].lower():
                print("%-72s %-14s %6d %16.1f %10.2f" % (short(r[0]), r[1], r[2], r[3], r[4] / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
