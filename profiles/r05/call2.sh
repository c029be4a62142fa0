#!/bin/bash
# GPU call 2 of round 5: the persistent launch of the pipelined encoder (device/pipe_persist.h) on the MI355X for the first time:
# parity (the encoder tests of the GPU suite go through it by default), then the headline both ways.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 400 python profiles/r05/persist_check.py 60 > $O/c2_check.log 2>&1
tail -8 $O/c2_check.log
timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c2_persist.json 2> $O/c2_persist.err
tail -c 900 $O/c2_persist.json; tail -3 $O/c2_persist.err
ZPAQ_AMD_PIPE_PERSIST=0 timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c2_classic.json 2> $O/c2_classic.err
tail -c 900 $O/c2_classic.json; tail -3 $O/c2_classic.err
for nb in 64 256 512; do
  timeout 200 python bench.py --blocks $nb --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c2_persist_$nb.json 2> $O/c2_persist_$nb.err
  python -c "import json,sys; d=json.load(open('$O/c2_persist_$nb.json')); print($nb, d['value'], d['persistent_launch'], d['kernel_ms'])"
done
