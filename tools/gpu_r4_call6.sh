#!/bin/bash
# round 4, call 6: complex fix, device hooks, spectral Laplacian (hipFFT), decomposed expression PDEs in the C loops (exchange to self) + whole suite
O=gpurun_out/r4f
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_complex.py tests/test_hip_frows.py tests/test_hip_operators.py tests/test_hip_distributed.py -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_new.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_new.log; grep "^FAILED\|^ERROR" $O/gpu_new.log | head -20
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_hip_complex.py --deselect tests/test_hip_frows.py --deselect tests/test_hip_operators.py --deselect tests/test_hip_distributed.py > $O/gpu_pytest.log 2>&1
echo "rc=$?"; tail -3 $O/gpu_pytest.log; grep "^FAILED\|^ERROR" $O/gpu_pytest.log | head -20
python - <<'PY'
# spectral operator timing at 4096^2 fp64 next to the finite-difference operator
import sys, time
sys.path[:0] = [".", "py-pde_amd"]
import numpy as np, pde_hip
from pde_hip.device import DeviceArray
b = pde_hip.get_backend("hip")
for n in (1024, 4096):
    grid = pde_hip.CartesianGrid([[0, n]] * 2, n, periodic=True)
    info = b.grid_info(grid, np.float64)
    a, o = DeviceArray(info).set_valid(np.random.default_rng(0).random(grid.shape)), DeviceArray(info)
    from pde_hip import _abi
    for name, call in (("spectral", lambda: b._lib.laplace_spectral(info.ref, a.ptr, o.ptr, _abi.OUT_FULL, b.stream)), ("stencil", lambda: b._lib.laplace(info.ref, a.ptr, o.ptr, _abi.OUT_FULL, b.stream))):
        for _ in range(3): call()
        b.synchronize(); t0 = time.perf_counter()
        for _ in range(20): call()
        b.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"laplace {name} {n}^2 fp64: {dt*1e3:.3f} ms  ({n*n*16/dt/1e9:.0f} GB/s of 16 B/cell)")
PY
