"""Per-kernel averages of rocprofv3 --pmc counters (rocpd sqlite) as a markdown table.

usage: python tools/rocprof_pmc_summary.py <dir-or-db> [<dir-or-db> ...] [-o out.md]
"""
import glob
import os
import sqlite3
import sys


def rows_of(src):
    dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
    for db in dbs:
        con = sqlite3.connect(db)
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration), max(scratch_size) "
             "from counters_collection group by kernel_name, counter_name order by sum(duration) desc")
        yield from con.execute(q)
        con.close()


def main():
    args = sys.argv[1:]
    out = None
    if "-o" in args:
        i = args.index("-o")
        out = args[i + 1]
        del args[i:i + 2]
    lines = ["| kernel | counter | dispatches | avg | min | max | avg duration (us) | scratch (B/lane) |", "|---|---|---:|---:|---:|---:|---:|---:|"]
    for src in args:
        for name, counter, n, avg, lo, hi, dur, scratch in rows_of(src):
            lines.append(f"| `{name[:140]}` | {counter} | {n} | {avg:.1f} | {lo:.1f} | {hi:.1f} | {dur / 1e3:.1f} | {scratch} |")
    text = "\n".join(lines)
    print(text)
    if out:
        with open(out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
