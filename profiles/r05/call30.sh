#!/bin/bash
# GPU call 30 of round 5: the eight-engine test again (call 29: out of memory -- the test process itself held most of the HBM from the
# tests before it, and eight engines on one device each budgeted 85 % of what was free) behind a test that fills the HBM, and the
# four tests call 29's -x did not reach
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --durations=8 -k "lockstep_decoder_on_the_mixed or eight_engines or persistent_launch_gives_up or input_tail or suffix_arrays or lz77_parse" > $O/c30_tests.txt 2>&1
tail -16 $O/c30_tests.txt
