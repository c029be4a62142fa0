#!/bin/bash
# GPU call 26 of round 5: the archiver's batch, per-unit profile of its SECOND (short) chain
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
ZPAQ_AMD_PERSIST_PROF_RUN=1 timeout 200 python profiles/r05/cli_bench.py --quick --out $O/c26_cli_quick.json > $O/c26_cli.log 2>&1
python -c "
import json
for r in json.load(open('$O/c26_cli_quick.json'))['rows']: print(r['what'][:60], round(r.get('wall_s',0),2), r.get('library_log'), r.get('archiver_says'))"
python profiles/persist_prof.py $O/cli_prof_1.bin | head -125
