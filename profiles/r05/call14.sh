#!/bin/bash
# GPU call 14 of round 5: ISSE pairs packed in LDS: 7 workgroups per group of the -m5 chain (224 of 256 CUs) against 8; the mixed corpus
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for w in 0 8; do
  ZPAQ_AMD_PERSIST_WPG_MIN=$w ZPAQ_AMD_PERSIST_PROF=$O/c14_prof_w$w.bin timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c14_w$w.json 2> $O/c14_w$w.err
  python -c "import json,sys; d=json.load(open('$O/c14_w$w.json')); print('wpg_min', $w, round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
  python profiles/persist_prof.py $O/c14_prof_w$w.bin > $O/c14_prof_w$w.txt
done
timeout 600 python bench.py --kind mixed --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c14_mixed.json 2> $O/c14_mixed.err
python -c "import json,sys; d=json.load(open('$O/c14_mixed.json')); print('mixed', round(d['value'],1), d['persistent_launch'], d['kernel_ms'], d['config']['ncomp'])"
tail -3 $O/c14_mixed.err
