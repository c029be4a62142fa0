#!/bin/bash
# GPU call 24 of round 5: kernel trace of the patched archiver's add: do its two chains' persistent launches overlap?
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 200 python profiles/r05/cli_bench.py --quick --keep --out $O/c24_cli_quick.json > $O/c24_cli.log 2>&1
cd /tmp/zpq_cli_bench && export TMPDIR=/tmp
ZPAQ_AMD_LOG=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/c24_trace -o t -- $R/oracle/_ref/zpaq_amd_cli_batch add /tmp/zpq_cli_bench/t.zpaq tree -method 50 -threads 4 > $O/c24_trace.log 2>&1
grep "zpaq_amd\]" $O/c24_trace.log
cd $R
python - <<PY
import csv, glob
for f in glob.glob("$O/c24_trace/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d > 5: print(r["Kernel_Name"][:50], "queue", r.get("Queue_Id"), "start %.1f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6), "dur %.1f ms" % d, "grid", r.get("Grid_Size_X"), "wg", r.get("Workgroup_Size_X"), "lds", r.get("LDS_Block_Size"))
PY
find $O -name "*.db" -delete 2>/dev/null
