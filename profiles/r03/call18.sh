#!/bin/bash
# call 18: the golden archives through the experimental decoder with two blocks per wavefront (wt_dual_tmp/)
mkdir -p gpurun_out/r03c18
cd wt_dual_tmp || { echo "no worktree copy"; exit 1; }
timeout 16 python profiles/r03/golden_dual.py > ../gpurun_out/r03c18/out.txt 2>&1
echo "rc=$?" >> ../gpurun_out/r03c18/out.txt
tail -5 ../gpurun_out/r03c18/out.txt
