#!/bin/bash
# GPU call 22 of round 5: the chain the archiver's text blocks get ("5,R,1": 25 components, two word models) in the latency shape, per-unit profile
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
ZPAQ_AMD_PERSIST_PROF=$O/c22_prof_text256.bin timeout 200 python bench.py --method 5,128,1 --blocks 256 --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c22_text256.json 2> $O/c22_text256.err
tail -3 $O/c22_text256.err
python -c "import json; d=json.load(open('$O/c22_text256.json')); print('text chain 256', round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
python profiles/persist_prof.py $O/c22_prof_text256.bin | head -130
