#!/bin/bash
# Round 3, GPU call 13: the C++ API / CLI tests after the last engine changes (blocks of several segments), kernel trace of the decoder
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c13
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 500 python -m pytest tests/test_cpp_api.py tests/test_cli.py -m gpu -q -x > $O/tests.txt 2>&1
tail -3 $O/tests.txt | tee $O/summary.txt
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --mode decode --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-200 | tee -a $O/summary.txt
find $O/prof -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
