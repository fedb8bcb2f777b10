#!/bin/bash
# round 5, call 35: last check of the in-tree library (comm TU rebuilt after two reverted experiments): distributed GPU tests + smoke()
cd /root/repo
timeout 900 python -m pytest tests/test_hip_distributed.py tests/test_hip_euler2.py -m gpu -q 2>&1 | grep -E "passed|failed"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
