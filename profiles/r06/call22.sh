#!/bin/bash
# GPU call 22 of round 6: the failing test's sequence of batches in one process (several chains one after the other)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
for v in "X=1 A=test" "X=1 A=long-first" "X=1 A=many-first" "ZPAQ_AMD_SMALL_CHAIN=0 A=test" "ZPAQ_AMD_SMALL_CHAIN=0 ZPAQ_AMD_CODER_FAST=0 A=test" "ZPAQ_AMD_STREAM_AHEAD=0 A=test" "ZPAQ_AMD_PIPE_PERSIST=0 A=test"; do
  echo "== $v"
  a=${v##*A=}
  env ${v% A=*} timeout 600 python profiles/r06/debug_seq.py $a 2>&1 | grep -v "amdgpu.ids"
done > $O/c22_debug.txt 2>&1
grep -v "^+" $O/c22_debug.txt
