#!/bin/bash
# GPU call 39 of round 6: rocprofv3 --kernel-trace --stats of the default line's command on the final code
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
timeout 800 rocprofv3 --kernel-trace --stats -d $O/c39_trace -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 --legacy 0 > $O/c39_trace_bench.json 2> $O/c39_trace_bench.err
echo "trace rc=$?"
cd $R
python - <<PY
import glob, json
for f in glob.glob("$O/c39_trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:800])
j = json.loads([l for l in open("$O/c39_trace_bench.json") if l.startswith("{")][-1])
print("traced line", round(j["value"], 1), "code ms", j["kernel_ms"]["code"], "configs1", (j.get("configs1") or {}).get("value"))
PY
find $O -name "*.db" -delete 2>/dev/null
