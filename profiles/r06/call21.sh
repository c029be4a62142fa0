#!/bin/bash
# GPU call 21 of round 6: the small-chain GPU test fails on the one-ICM chain (emulator passes): which blocks, under which knobs
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
for v in "X=1" "ZPAQ_AMD_CODER_FAST=0" "ZPAQ_AMD_STREAM_AHEAD=0" "ZPAQ_AMD_SMALL_CHAIN=0" "ZPAQ_AMD_PIPE_PERSIST=0"; do
  echo "== $v"
  env $v timeout 600 python profiles/r06/debug_small.py 2>&1 | grep -v "amdgpu.ids"
done > $O/c21_debug.txt 2>&1
cat $O/c21_debug.txt
