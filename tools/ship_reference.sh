#!/bin/bash
# Copy the reference's package and tests into the git-ignored scratch directory `_refscratch/` so that ONE gpurun call can
# run the drop-in tests against the real libpdehip.so on the MI355X (tools/gpu_r3_dropin.sh); `rm` removes it again.
# The copy is the CHECKER (like tests/golden/make_golden.py imports it here); it is never committed (.gitignore) and is
# removed right after the call.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "rm" ]; then rm -rf _refscratch; echo "removed _refscratch"; exit 0; fi
rm -rf _refscratch; mkdir -p _refscratch
cp -r /root/reference/pde _refscratch/pde
cp -r /root/reference/tests _refscratch/tests
find _refscratch -name "__pycache__" -type d -prune -exec rm -rf {} +
du -sh _refscratch
