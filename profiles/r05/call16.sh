#!/bin/bash
# GPU call 16 of round 5: the whole GPU suite on the round's code, the default bench line, configs[1], the archiver (patched / unpatched)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/c16_gputest.txt 2>&1
tail -25 $O/c16_gputest.txt
timeout 900 python bench.py > $O/c16_bench_default.json 2> $O/c16_bench_default.err
tail -c 2500 $O/c16_bench_default.json; tail -3 $O/c16_bench_default.err
timeout 300 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 > $O/c16_configs1.json 2> $O/c16_configs1.err
python -c "import json; d=json.load(open('$O/c16_configs1.json')); print('configs1', round(d['value'],1), d['persistent_launch'], d['kernel_ms'], (d.get('api') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value'))"
timeout 900 python profiles/r05/cli_bench.py --files 256 --out $O/c16_cli.json > $O/c16_cli.log 2>&1
cat $O/c16_cli.log | cut -c1-250
