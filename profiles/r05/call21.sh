#!/bin/bash
# GPU call 21 of round 5: the patched archiver again with the library saying which launch form coded the batch
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 200 python profiles/r05/cli_bench.py --quick --out $O/c21_cli_quick.json > $O/c21_cli.log 2>&1
python -c "
import json
for r in json.load(open('$O/c21_cli_quick.json'))['rows']: print(r['what'][:60], round(r.get('wall_s',0),2), r.get('library_log'), r.get('archiver_says'))"
