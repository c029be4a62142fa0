#!/bin/bash
# GPU call 27 of round 5: which XCDs does the short run of the archiver's batch have to be rotated to?  (the add, 8 rotations)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 100 python profiles/r05/cli_bench.py --quick --keep --out $O/c27_cli_quick.json > $O/c27_cli.log 2>&1
cd /tmp/zpq_cli_bench
for k in 0 1 2 3 4 5 6 7; do
  rm -f t.zpaq
  ZPAQ_AMD_PERSIST_ROT=$k ZPAQ_AMD_LOG=1 timeout 60 $R/oracle/_ref/zpaq_amd_cli_batch add /tmp/zpq_cli_bench/t.zpaq tree -method 50 -threads 4 2>&1 | grep -o "rot.*\|coding [0-9.]* on" | tr '\n' ' '; echo " <- extra rotation $k"
done
