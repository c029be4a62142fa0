#!/bin/bash
# GPU call 25 of round 5: runs of one batch placed per XCD (short run rotated to the emptiest XCDs): the archiver's batch again,
# the mixed corpus (two chains that fill the device exactly), small batches with a group per XCD against a group over all XCDs
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 200 python profiles/r05/cli_bench.py --quick --out $O/c25_cli_quick.json > $O/c25_cli.log 2>&1
python -c "
import json
for r in json.load(open('$O/c25_cli_quick.json'))['rows']: print(r['what'][:60], round(r.get('wall_s',0),2), r.get('library_log'), r.get('archiver_says'))"
timeout 300 python bench.py --kind mixed --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c25_mixed.json 2> $O/c25_mixed.err
python -c "import json; d=json.load(open('$O/c25_mixed.json')); print('mixed 1024', round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
for sp in 1 8; do for nb in 64 128; do
  ZPAQ_AMD_PERSIST_SPREAD=$sp timeout 200 python bench.py --blocks $nb --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c25_b${nb}_s$sp.json 2> $O/c25_b${nb}_s$sp.err
  python -c "import json; d=json.load(open('$O/c25_b${nb}_s$sp.json')); print('spread', $sp, 'blocks', $nb, round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
done; done
