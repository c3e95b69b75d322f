#!/usr/bin/env python3
"""print the per-kernel summary (calls, total us, average us, %) of a rocprofv3 results database (rocpd sqlite)"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
print("%-70s %6s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print("%-70s %6d %12.1f %10.2f %6.1f" % (name[:70], calls, total, avg, pct))
