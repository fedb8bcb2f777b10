"""Tuning aid: which fused output-ghost faces cost time?  (spawns one process per setting)"""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
n = sys.argv[1] if len(sys.argv) > 1 else "512"
for tune in ["4,4,1,1,512", "2,4,1,1,1024"]:
    for axes in ["0", "1", "2", "4", "7"]:
        env = dict(os.environ, PDEHIP_TUNE=tune, PDEHIP_FUSE_AXES=axes)
        print(f"fuse_axes={axes} ", end="", flush=True)
        subprocess.run([sys.executable, str(HERE / "sweep.py"), n, "worker"], env=env, check=False)
