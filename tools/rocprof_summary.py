"""Dump the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite) as text.

usage: python tools/rocprof_summary.py <dir-or-db> [out.md]
"""
import glob
import os
import sqlite3
import sys


def main():
    src = sys.argv[1]
    dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
    lines = []
    for db in dbs:
        con = sqlite3.connect(db)
        lines.append(f"## {os.path.basename(db)}")
        lines.append("| kernel | calls | total (us) | avg (us) | % |")
        lines.append("|---|---:|---:|---:|---:|")
        for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            lines.append(f"| `{name[:150]}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |")
        con.close()
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
