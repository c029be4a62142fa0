#!/bin/bash
# GPU call 15 of round 5: headline and mixed corpus with 8 workgroups per group for both chains (256 workgroups), parity check
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c15_text.json 2> $O/c15_text.err
python -c "import json,sys; d=json.load(open('$O/c15_text.json')); print('text', round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
timeout 600 python bench.py --kind mixed --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c15_mixed.json 2> $O/c15_mixed.err
python -c "import json,sys; d=json.load(open('$O/c15_mixed.json')); print('mixed', round(d['value'],1), d['persistent_launch'], d['kernel_ms'], d['config']['ncomp'], d['all_status_ok'], d['roundtrip_verified_blocks'])"
tail -3 $O/c15_mixed.err
timeout 400 python profiles/r05/persist_check.py 60 > $O/c15_check.log 2>&1
tail -6 $O/c15_check.log
