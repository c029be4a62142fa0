#!/bin/bash
# GPU call 28 of round 5: exact grids (whole sets of 8 groups one XCD each, the rest one after the other), runs of one batch sized
# per XCD: the archiver's batch, the mixed corpus, the headline, the GPU tests of the pipelined encoder
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 200 python profiles/r05/cli_bench.py --quick --out $O/c28_cli_quick.json > $O/c28_cli.log 2>&1
python -c "
import json
for r in json.load(open('$O/c28_cli_quick.json'))['rows']: print(r['what'][:60], round(r.get('wall_s',0),2), r.get('library_log'), r.get('archiver_says'))"
timeout 300 python bench.py --kind mixed --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c28_mixed.json 2> $O/c28_mixed.err
python -c "import json; d=json.load(open('$O/c28_mixed.json')); print('mixed 1024', round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c28_headline.json 2> $O/c28_headline.err
python -c "import json; d=json.load(open('$O/c28_headline.json')); print('headline', round(d['value'],1), d['persistent_launch'], d['kernel_ms'], d['reference_identity']['identical'])"
timeout 200 python bench.py --blocks 448 --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c28_b448.json 2> $O/c28_b448.err
python -c "import json; d=json.load(open('$O/c28_b448.json')); print('448 blocks (14 groups)', round(d['value'],1), d['persistent_launch'], d['kernel_ms'], d['reference_identity']['identical'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "both_shapes_of_the_pipelined or persistent_launch_gives_up or mixed_corpus_batch or mixed_plans or two_engines" > $O/c28_tests.txt 2>&1; tail -3 $O/c28_tests.txt
