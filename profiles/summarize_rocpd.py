#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd .db output) as text.

usage: python profiles/summarize_rocpd.py <results.db> [> profiles/rNN_kernel_stats.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("# source: %s  (rocprofv3 --kernel-trace --stats; durations in microseconds)" % sys.argv[1])
print("%-110s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-110s %8d %14.3f %12.3f %8.2f" % (name[:110], calls, total / 1e3 if total > 1e6 else total, avg / 1e3 if total > 1e6 else avg, pct))
cur = db.execute("select name, count(*), avg(duration), min(duration), max(duration), max(lds_size), max(vgpr_count), "
                 "max(sgpr_count), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc")
print()
print("%-60s %6s %12s %12s %12s %8s %5s %5s %7s %8s %5s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "lds", "vgpr", "sgpr", "scratch", "grid_x", "wg_x"))
for r in cur.fetchall():
    print("%-60s %6d %12.0f %12d %12d %8d %5d %5d %7d %8d %5d" % (r[0][:60], *r[1:]))
