import os, subprocess, sys
from pathlib import Path
HERE = Path(__file__).resolve().parent
for cfg in ["default", "2,2,1,1,1024", "2,2,1,1,2048", "2,2,1,1,4096", "4,2,1,1,1024", "4,2,1,1,2048", "4,2,1,1,4096", "2,1,1,1,2048", "2,1,1,1,4096"]:
    env = dict(os.environ, SWEEP_DTYPE="float32")
    if cfg != "default":
        env["PDEHIP_TUNE"] = cfg
    subprocess.run([sys.executable, str(HERE / "sweep.py"), sys.argv[1] if len(sys.argv) > 1 else "512", "worker"], env=env, check=False)
