#!/bin/bash
# GPU call 27 of round 4: the walk kernel with literal runs in one step -- parity tests again, configs[1], kernel trace.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
(time timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lz77_parse or suffix_arrays") > $O/gputest_lz77b.txt 2>&1
tail -5 $O/gputest_lz77b.txt
(time timeout 300 python bench.py --method 3 --blocks 256 --block-bytes 262144 --kind lcg --decode-blocks 0 --steps 3) > $O/bench_configs1_final2.json 2> $O/bench_configs1_final2.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_configs1_final2.json", errors="replace") if l.startswith("{")][-1])
print("configs[1] value", round(d["value"], 1), "api", round(d["api"]["value"], 1), d["api"]["ms"]["host_front"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("bit_identical_vs_reference"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_c1b -o p --output-format csv -- python $R/bench.py --method 3 --blocks 256 --block-bytes 262144 --kind lcg --decode-blocks 0 --steps 1 --warmup 0 --cpu-seconds 0 > $O/prof_c1b.log 2>&1
find $O/prof_c1b -name "*kernel_stats.csv" -exec grep -E "lz77|Name" {} \; | cut -c1-200
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
