#!/bin/bash
# A/B matrix of the experiment variants in ONE gpurun call (build them first: python profiles/ab_build.py).
#   gpurun --timeout 600 -- 'bash profiles/ab_run.sh > gpurun_out/ab.txt 2>&1; tail -40 gpurun_out/ab.txt'
# Prints one line per run: variant, blocks x bytes, kernel ms, kernel MB/s.  64 KiB blocks keep a run at ~1-2 s.
run() {  # env-assignments  blocks  bytes
  env $1 timeout 150 python bench.py --blocks $2 --block-bytes $3 --cpu-seconds 0 --warmup 0 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    j = json.loads(l)
    ms = j['kernel_ms']['code']
    print('%-58s %5d x %-8d code_ms=%9.1f  kernel MB/s=%7.2f  ok=%s' % ('$1', $2, $3, ms, $2 * $3 / 1e3 / ms, j['all_status_ok']))
except Exception as e:
    print('%-58s %5d x %-8d FAILED %s' % ('$1', $2, $3, l[-200:]))
"
}
rund() {  # dual-wavefronts  blocks  bytes : the experimental two-blocks-per-wavefront kernel (bench.py --dual)
  timeout 150 python bench.py --dual $1 --blocks $2 --block-bytes $3 --cpu-seconds 0 --warmup 0 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    j = json.loads(l)
    ms = j['kernel_ms']['code']
    print('%-58s %5d x %-8d code_ms=%9.1f  kernel MB/s=%7.2f  ok=%s verified=%s' % ('--dual $1', $2, $3, ms, $2 * $3 / 1e3 / ms, j['all_status_ok'], j['roundtrip_verified_blocks']))
except Exception as e:
    print('%-58s %5d x %-8d FAILED %s' % ('--dual $1', $2, $3, l[-300:]))
"
}
T="ZPAQ_AMD_SPEC_DEFS=-DZPQ_TOUCH2=1"
for size in 65536 1048576; do
  run "ZPAQ_AMD_SPEC_WAVES=4" 1024 $size
  run "ZPAQ_AMD_SPEC_WAVES=4 $T" 1024 $size
  run "ZPAQ_AMD_SPEC_WAVES=8" 2048 $size
  run "ZPAQ_AMD_SPEC_WAVES=8 $T" 2048 $size
  if [ $size = 65536 ]; then
    # -m5 keeps 97 MiB of state per block whatever the block size: at most ~2 500 blocks are resident in 288 GB,
    # so the 12-block shape cannot be filled completely (204 of 256 workgroup slots) and the 16-block shape not at all
    run "ZPAQ_AMD_SPEC_WAVES=12" 2448 $size
    run "ZPAQ_AMD_SPEC_WAVES=12 $T" 2448 $size
    rund 4 2048 $size      # 8 blocks per CU, one wavefront per SIMD
    rund 8 2048 $size      # 8 blocks per CU on half the CUs ... and
    rund 8 2400 $size      # ... as many as HBM holds
  fi
done
