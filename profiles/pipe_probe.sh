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
run "ZPAQ_AMD_PIPE_SLACK=0" 1024 65536
run "ZPAQ_AMD_PIPE_SLACK=1" 1024 65536
run "ZPAQ_AMD_PIPE_SLACK=6" 1024 65536
run "ZPAQ_AMD_PIPE_CHUNK=1024" 1024 65536
run "ZPAQ_AMD_PIPE_CHUNK=256 ZPAQ_AMD_PIPE_SLACK=6" 1024 65536
run "X=1" 1024 1048576
run "X=1" 2048 1048576
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --blocks 1024 --block-bytes 65536 --cpu-seconds 0 --warmup 0 --api-blocks 0 --verify-blocks 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/pipe_timeline.py gpurun_out/prof_tl/tl_results.db
