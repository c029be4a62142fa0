#!/bin/bash
# GPU call 31 of round 5: the eight-engine test with chains the build knows (no hipRTC)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --durations=4 -k "eight_engines or two_engines" > $O/c31_tests.txt 2>&1
tail -8 $O/c31_tests.txt
