#!/bin/bash
# GPU call 10 of round 4 (final): the whole GPU suite, the default bench line as the driver runs it, and the rocprofv3 kernel
# traces of the same commands (headline encoder; decoder at the configs[4] operating point).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
(time timeout 900 python -m pytest tests -x -q -m gpu) > $O/gputest.txt 2>&1
tail -6 $O/gputest.txt
(time timeout 420 python bench.py) > $O/bench_final.json 2> $O/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_final.json", errors="replace") if l.startswith("{")][-1])
print("headline", round(d["value"], 1), "frac", round(d["roofline"]["frac"], 4), "api", round(d["api"]["value"], 1), d["api"]["ms"])
print("decode", {k: d["decode"].get(k) for k in ("value", "ms", "every_byte_verified")}, d["decode"]["roofline"]["kernel"], round(d["decode"]["roofline"]["frac"], 4))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("bit_identical_vs_reference"), "cpu decode", d["decode"]["cpu_baseline"]["value"] if d["decode"].get("cpu_baseline") else None)
PY
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_enc -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 --api-blocks 0 --decode-blocks 0 --verify-blocks 0 --warmup 0 --steps 1 > $O/prof_enc.log 2>&1
find $O/prof_enc -name "*kernel_stats.csv" -exec head -12 {} \;
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dec -o p --output-format csv -- python $R/bench.py --mode decode --cpu-seconds 0 --verify-blocks 0 --warmup 0 --steps 1 > $O/prof_dec.log 2>&1
find $O/prof_dec -name "*kernel_stats.csv" -exec head -6 {} \;
tail -c 600 $O/prof_dec.log
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
