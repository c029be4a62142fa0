#!/bin/bash
run() {  # env blocks bytes
  env $1 timeout 300 python bench.py --blocks $2 --block-bytes $3 --cpu-seconds 0 --warmup 1 --verify-blocks 2 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    j = json.loads(l)
    ms = j['kernel_ms']['code']
    print('%-58s %5d x %-8d code_ms=%9.1f  kernel MB/s=%8.2f  ok=%s verified=%s' % ('$1', $2, $3, ms, $2 * $3 / 1e3 / ms, j['all_status_ok'], j['roundtrip_verified_blocks']))
except Exception as e:
    print('%-58s %5d x %-8d FAILED %s' % ('$1', $2, $3, l[-300:]))
"
}
python profiles/pipe_probe.py
run "X=1" 1024 65536
run "ZPAQ_AMD_PIPE_GROUP=64" 1024 65536
run "ZPAQ_AMD_PIPE_GROUP=16" 1024 65536
run "X=1" 2048 65536
run "X=1" 1024 1048576
ZPAQ_AMD_PIPE_PROFILE=1 python bench.py --blocks 1024 --block-bytes 65536 --cpu-seconds 0 --warmup 0 --verify-blocks 0 2>&1 | grep "pipe profile" | awk '{k=$4; t[k]+=$7; n[k]+=1; if ($7>m[k]) m[k]=$7} END {for (k in t) printf "alone: %-6s units=%d avg=%.3f max=%.3f ms/step\n", k, n[k], t[k]/n[k], m[k]}'
