#!/bin/bash
# GPU call 23 of round 4 (final code): the decoder's GPU tests, the default bench line, the decoder's kernel trace.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
(time timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lockstep or two_blocks or decode or status or large_batch or specialised or cross_lane") > $O/gputest_decoders_final23.txt 2>&1
tail -4 $O/gputest_decoders_final23.txt
(time timeout 420 python bench.py) > $O/bench_final23.json 2> $O/bench_final23.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_final23.json", errors="replace") if l.startswith("{")][-1])
print("headline", round(d["value"], 1), "frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], "api", round(d["api"]["value"], 1), d["api"]["ms"]["library_total"])
print("decode", {k: d["decode"].get(k) for k in ("value", "ms", "every_byte_verified")}, d["decode"]["roofline"]["kernel"], round(d["decode"]["roofline"]["frac"], 4))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("bit_identical_vs_reference"), "cpu decode", d["decode"]["cpu_baseline"]["value"] if d["decode"].get("cpu_baseline") else None)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dec23 -o p --output-format csv -- python $R/bench.py --mode decode --cpu-seconds 0 --verify-blocks 0 --warmup 0 --steps 1 > $O/prof_dec23.log 2>&1
find $O/prof_dec23 -name "*kernel_stats.csv" -exec head -4 {} \;
find $O -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
