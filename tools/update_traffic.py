"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-extra`: HBM bytes per launch of the dominant
kernel, keyed by the INSTANCE name `pdehip_last_kernel_name` reports in the bench line (bench.py looks the entry up by that name, so a profile
of another instance never labels a run).  FETCH_SIZE is doubled (gfx950: MI355X_MICROARCH.md, HBM section: the counter tallies 128-byte requests
of wide coalesced reads at 64 bytes); both counters are in kilobytes... see `unit`.

usage: python tools/update_traffic.py <fetch-dir> <write-dir> <bench.json of the same build> <source-note>
"""
import glob
import json
import os
import sqlite3
import sys

fetch_dir, write_dir, bench_json, source = sys.argv[1:5]


def per_kernel(src, counter):
    db = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[0]
    con = sqlite3.connect(db)
    rows = list(con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name order by sum(duration) desc", (counter,)))
    con.close()
    return rows


fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
line = json.loads(open(bench_json).read().strip().splitlines()[-1])
instance = line["roofline"]["kernel"].split(" - ")[0]
n = round(line["config"]["cells"] ** (1 / 3))
# the profiled symbol of the instance: its template name (before `<`) inside the kernel name, the one with the most dispatches
stem = instance.split("<")[0]
cands = [r for r in fetch if stem in r[0]]
if not cands:
    sys.exit(f"no kernel named like {stem} in {fetch_dir}")
name, nf, f = max(cands, key=lambda r: r[1])
wmap = {k: v for k, _, v in write}
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes (x 1024 B); FETCH_SIZE x 2 on gfx950
bytes_per_launch = int(round((2.0 * f + wmap.get(name, 0.0)) * 1024))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")
try:
    tj = json.load(open(path))
except (OSError, ValueError):
    tj = {}
tj.setdefault("kernels", {})[f"{instance} @ {n}^3"] = {"bytes_per_launch": bytes_per_launch, "fetch_kb_avg": f, "write_kb_avg": wmap.get(name), "dispatches": nf,
                                                      "profiled_symbol": name[:160], "source": source}
tj["_comment"] = "HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, kilobytes x 1024); `kernels` is keyed by the instance name the library reports"
json.dump(tj, open(path, "w"), indent=1)
print(instance, "@", n, "->", bytes_per_launch, "bytes per launch; moved bytes", 16 * n**3, "ratio", round(bytes_per_launch / (16 * n**3), 4))
