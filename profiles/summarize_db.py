#!/usr/bin/env python3
"""Dump the per-kernel summary (calls, total/avg duration in us, share) of a
rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats` default output on
ROCm 7.2) as CSV, so the numbers can be committed under profiles/."""
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, "w") as f:
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for name, calls, total, avg, pct in rows:
            if len(name) > 160:
                name = name[:157] + "..."
            f.write('"%s",%d,%.3f,%.3f,%.2f\n' % (name.replace('"', "'"), calls, total, avg, pct))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
