#!/bin/bash
# round 5, nineteenth GPU session: the suite with the new tests (exhaustive reciprocal, lane region on a synchronous context), Sim3 timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s19
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed" $O/pytest.log | tail -3
timeout 200 python tools/bench_sim3.py 2>&1 | tail -2 | tee $O/sim3.json
