#!/bin/bash
# GPU call 29 of round 5: the final code -- the whole GPU suite, the default bench line as the driver runs it, its kernel trace,
# the other configurations, the archiver
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q --durations=12 > $O/c29_gputest.txt 2>&1
tail -20 $O/c29_gputest.txt
timeout 600 python bench.py > $O/c29_bench_default.json 2> $O/c29_bench_default.err
tail -c 1500 $O/c29_bench_default.json; tail -3 $O/c29_bench_default.err
timeout 200 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 > $O/c29_bench_configs1.json 2> $O/c29_bench_configs1.err
timeout 300 python bench.py --kind mixed --decode-blocks 0 --api-blocks 0 > $O/c29_bench_mixed.json 2> $O/c29_bench_mixed.err
timeout 300 python bench.py --blocks 2048 --decode-blocks 0 --api-blocks 0 --cpu-seconds 0 > $O/c29_bench_dense.json 2> $O/c29_bench_dense.err
for f in configs1 mixed dense; do python -c "import json; d=json.load(open('$O/c29_bench_$f.json')); print('$f', round(d['value'],1), d['persistent_launch'], d['kernel_ms'], (d.get('api') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value'), (d.get('reference_identity') or {}).get('identical'))"; done
timeout 400 python profiles/r05/cli_bench.py --files 256 --skip-unpatched-extract --out $O/c29_cli.json > $O/c29_cli.log 2>&1
cut -c1-220 $O/c29_cli.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c29_trace -o t -- python $R/bench.py --cpu-seconds 0 > $O/c29_trace_bench.json 2> $O/c29_trace.err
cd $R
ls $O/c29_trace/ | head; head -12 $O/c29_trace/*kernel_stats.csv 2>/dev/null | cut -c1-200
find $O -name "*.db" -delete 2>/dev/null
find $O/c29_trace -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
