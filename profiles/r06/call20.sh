#!/bin/bash
# GPU call 20 of round 6: the small-chain GPU test failed on its first batch (ragged blocks) where the emulator passes: which
# blocks, and under which knobs
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
for v in "X=1" "ZPAQ_AMD_CODER_FAST=0" "ZPAQ_AMD_STREAM_AHEAD=0" "ZPAQ_AMD_SMALL_CHAIN=0" "ZPAQ_AMD_SMALL_CHAIN=0 ZPAQ_AMD_CODER_FAST=0" "ZPAQ_AMD_PIPE_PERSIST=0"; do
  echo "== $v"
  env $v timeout 400 python profiles/r06/debug_small.py 2>&1 | grep -v amdgpu.ids | tail -9
done > $O/c20_debug.txt 2>&1
cat $O/c20_debug.txt
