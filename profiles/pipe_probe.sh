#!/bin/bash
run() {  # env blocks bytes
  env $1 timeout 300 python bench.py --blocks $2 --block-bytes $3 --cpu-seconds 0 --warmup 1 --verify-blocks 2 --api-blocks 0 2>&1 | tail -1 | python -c "
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
run "ZPAQ_AMD_PIPE_MIX_SPLIT=2" 1024 65536
run "ZPAQ_AMD_PIPE_MIX_SPLIT=2 ZPAQ_AMD_PIPE_GROUP=64" 1024 65536
run "ZPAQ_AMD_PIPE_GROUP=16" 1024 65536
run "X=1" 1024 1048576
python -m pytest tests/test_cli.py tests/test_cpp_api.py -m gpu -q 2>&1 | tail -3
