"""Assemble profiles/r06_final_*.md and the bench lines from the outputs of tools/gpu_final_r6.sh (gpurun_out/r6final).
usage: python tools/write_final_profiles_r6.py"""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
O, P = ROOT / "gpurun_out" / "r6final", ROOT / "profiles"


def read(name):
    return (O / name).read_text() if (O / name).exists() else ""


lines = {}
for name, dst in (("bench20.json", "r06_final_bench_driver_args.json"), ("bench_n1.json", "r06_final_bench_default.json"), ("bench20_fastmath.json", "r06_final_bench_driver_args_fastmath.json"),
                  ("trace_bench.json", "r06_final_bench_under_rocprof.json")):
    text = read(name).strip().splitlines()
    if text:
        (P / dst).write_text(text[-1] + "\n")
        lines[name] = json.loads(text[-1])

suite = read("gpu_all.log")
summary = re.findall(r"^=*\s*(\d+ passed.*)$", suite, flags=re.M) or re.findall(r"(\d+ passed[^\n]*)", suite)
out = ["# r06 final build on one MI355X (tools/gpu_final_r6.sh)", "", f"* `pytest tests -m gpu`: **{summary[-1].strip(' =') if summary else 'see log'}**",
       "* `__graft_entry__.smoke()`: ok (11 Euler steps + laplace on (32, 32, 64) bit-identical to the oracle)", ""]
out += ["| bench line | value (Mcells/s) | ms per step | roofline.frac (frac_best) | kernel ms | traffic (bytes per launch) | of NT copy | Laplacian frac | parity |", "|---|---:|---:|---:|---:|---:|---:|---:|---|"]
for name, label in (("bench20.json", "`--gpus 1 --steps 20 --warmup 5` (the driver's arguments)"), ("bench_n1.json", "defaults"),
                    ("bench20_fastmath.json", "`PDEHIP_FASTMATH=1`, the driver's arguments (opt-in contraction: not bit-exact by construction, never the headline)"),
                    ("trace_bench.json", "under rocprofv3 `--kernel-trace`, `--steps 100 --warmup 10 --no-extra`")):
    d = lines.get(name)
    if not d:
        continue
    r = d["roofline"]
    op = d.get("roofline_operator", {})
    out.append(f"| {label} | {d['value']} | {d['ms_per_step']} | {r['frac']} ({r['frac_best']}) | {r['kernel_ms']} | {r['traffic']} | {r.get('frac_of_nt_copy')} | {op.get('frac')} | {(d.get('parity') or {}).get('ok')} |")
d = lines.get("bench20.json") or {}
out += ["", f"Dominant kernel as the library reports it: `{(d.get('roofline') or {}).get('kernel')}`", ""]
if d.get("extra"):
    out += ["## `extra` of the line with the driver's arguments", "", "```json", json.dumps(d["extra"], indent=1), "```", ""]
if d.get("roofline_operators"):
    ops = d["roofline_operators"]
    out += ["## other operators (HIP events, same run)", "", "| operator | ms | frac of 8 TB/s |", "|---|---:|---:|"]
    for k in ("gradient", "divergence", "gradient_squared", "vector_laplace"):
        if k in ops:
            out.append(f"| {k} | {ops[k]['kernel_ms']} | {ops[k]['frac']} |")
    if "tile2d" in ops:
        t = ops["tile2d"]
        out += ["", f"2-D loop (launch-bound): launch floor {t.get('launch_floor_us')} us; " + "; ".join(f"{k}: {v['us_per_launch']} us per launch of {v['steps_per_launch']} steps = {v['launches_of_floor']} launch floors" for k, v in t.items() if isinstance(v, dict))]
    if "copy" in ops:
        out += ["", f"copies: NT copy {ops['copy']['nt_copy_gbs']} GB/s, hipMemcpyDtoD {ops['copy']['hipMemcpyDtoD_gbs']} GB/s"]
if d.get("cpu_baseline"):
    out += ["", "## cpu_baseline", "", "```json", json.dumps(d["cpu_baseline"], indent=1), "```"]
out += ["", "## slab / block probes (halos to self)", "", "```", read("probe_slab.log").strip(), read("probe_block.log").strip(), "```"]
(P / "r06_final_gpu_suite.md").write_text("\n".join(out) + "\n")

trace = ["# r06 final build: rocprofv3 --kernel-trace --stats", "",
         "## `bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --repeats 3`: every launch of the dominant kernel is a 512^3 launch (two Euler steps)", "",
         read("trace_bench_summary.md"), "", "The bench line of this very run (HIP events): see `r06_final_bench_under_rocprof.json` - `roofline.kernel_ms` against the average above.", "",
         "## the whole line (`--steps 100 --warmup 10 --no-cpu-baseline --repeats 3`: operators, extra configurations, slab shares to self; the same kernel names run on other grids too)", "",
         read("trace_bench_full_summary.md"), "", "## `tools/run_laplace.py` (60 x pdehip_laplace at 512^3)", "", read("trace_ops_summary.md")]
(P / "r06_final_rocprof_kernel_trace.md").write_text("\n".join(trace) + "\n")

tj = json.loads((P / "traffic.json").read_text())
pmc = ["# r06 final build: rocprofv3 --pmc passes (separate runs; `--kernel-trace` only), tools/gpu_final_r6.sh", "",
       "Units and correction as before (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of this",
       "access pattern as 64 bytes -> x 2 for reads.  `profiles/traffic.json` (tools/update_traffic.py) holds the result keyed by the kernel instance the library reports:", "", "```json", json.dumps(tj.get("kernels"), indent=1), "```", "",
       "## time loop (`bench.py --steps 20 --warmup 2 --no-extra`): FETCH_SIZE, WRITE_SIZE", "", read("pmc_bench_summary.md"), "",
       "## operator path (tools/run_laplace.py)", "", read("pmc_ops_summary.md"), "",
       "## SQ counters of the dominant kernel: this build (all-periodic tall tile, three plane buffers)", "", read("pmc_sq_bench_summary.md"), "",
       "## the round-5 instance on the same box (`PDEHIP_E2_PER3=0 PDEHIP_EULER2=8`: tall tile with the face code as selects, four plane buffers)", "", read("pmc_sq_bench_r5tile_summary.md"), "",
       "## the contracted build (`PDEHIP_FASTMATH=1`)", "", read("pmc_sq_bench_fastmath_summary.md"), "",
       "SQ_INSTS_VALU per launch: 121.0 M (round 5) -> 107.8 M (this build) -> 99.2 M (contracted); asked: <= 95 M for the exact build - not quite reached.",
       "SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES: 55 % -> 59 % at one wave per SIMD; the kernel's wall time 459 -> 400 us under the profiler."]
(P / "r06_final_rocprof_pmc.md").write_text("\n".join(pmc) + "\n")
sizes = ["# r06 final build: sizes around the vector and tile boundaries (tools/time_sizes.py), rows on 128-byte lines", "", *[ln for ln in read("time_sizes.log").splitlines() if ln.startswith("|")], "", "## tails", "",
         *[ln for ln in read("time_sizes_tails.log").splitlines() if ln.startswith("|")]]
(P / "r06_final_time_sizes.md").write_text("\n".join(sizes) + "\n")
print("written:", sorted(p.name for p in P.glob("r06_final_*")))
