#!/bin/bash
# call 17: A/B of the experimental decoder with two blocks per wavefront (branch decoder-dual, copied to wt_dual_tmp/)
mkdir -p gpurun_out/r03c17
cd wt_dual_tmp || { echo "no worktree copy"; exit 1; }
timeout 80 python profiles/r03/ab_decode_dual.py ${1:-1280} 131072 > ../gpurun_out/r03c17/out.txt 2>&1
echo "rc=$?" >> ../gpurun_out/r03c17/out.txt
tail -20 ../gpurun_out/r03c17/out.txt
