#!/bin/bash
# GPU call 13 of round 5: persistent launch, which shape for how many blocks (latency: units per bit position; throughput: per block), 2048 blocks in two rounds
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for nb in 128 384 512 640 768 1024; do
  for mode in latency throughput; do
    ZPAQ_AMD_PIPE_MODE=$mode timeout 200 python bench.py --blocks $nb --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c13_${mode}_$nb.json 2> $O/c13_${mode}_$nb.err
    python -c "import json,sys; d=json.load(open('$O/c13_${mode}_$nb.json')); print('$mode', $nb, round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
  done
done
timeout 300 python bench.py --blocks 2048 --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c13_2048.json 2> $O/c13_2048.err
python -c "import json,sys; d=json.load(open('$O/c13_2048.json')); print('default', 2048, round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
