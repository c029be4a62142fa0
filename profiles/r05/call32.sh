#!/bin/bash
# GPU call 32 of round 5: the whole GPU suite in one run on the final code (call 29's run stopped at the eight-engine test)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 470 python -m pytest tests -m gpu -q --durations=10 > $O/c32_gputest.txt 2>&1
tail -18 $O/c32_gputest.txt
