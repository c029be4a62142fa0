#!/bin/bash
# GPU call 26 of round 4: configs[1] with the reference beside it (the line for profiles/) and the kernel trace of the same
# command (the pre-processor kernels' share).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
(time timeout 300 python bench.py --method 3 --blocks 256 --block-bytes 262144 --kind lcg --decode-blocks 0 --steps 3) > $O/bench_configs1_final.json 2> $O/bench_configs1_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_configs1_final.json", errors="replace") if l.startswith("{")][-1])
print("configs[1] value", round(d["value"], 1), "api", round(d["api"]["value"], 1), d["api"]["ms"]["host_front"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("bit_identical_vs_reference"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o p --output-format csv -- python $R/bench.py --method 3 --blocks 256 --block-bytes 262144 --kind lcg --decode-blocks 0 --steps 1 --warmup 0 --cpu-seconds 0 > $O/prof_c1.log 2>&1
find $O/prof_c1 -name "*kernel_stats.csv" -exec head -24 {} \;
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
