#!/bin/bash
# GPU call 10 of round 5: where the unit wavefronts of the persistent launch spend their time (ZPAQ_AMD_PERSIST_PROF)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for nb in 1024 256; do
  ZPAQ_AMD_PERSIST_PROF=$O/c10_prof_$nb.bin timeout 300 python bench.py --blocks $nb --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c10_$nb.json 2> $O/c10_$nb.err
  python -c "import json,sys; d=json.load(open('$O/c10_$nb.json')); print($nb, d['value'], d['persistent_launch'], d['kernel_ms'])"
  python profiles/persist_prof.py $O/c10_prof_$nb.bin > $O/c10_prof_$nb.txt
  cat $O/c10_prof_$nb.txt
done
