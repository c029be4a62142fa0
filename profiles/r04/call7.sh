#!/bin/bash
# GPU call 7 of round 4: the default bench line as the driver runs it (headline + API leg with the split copy + decode leg
# on the lockstep decoder + both CPU baselines), then the API leg alone with one copy up front (A/B).
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
(time timeout 420 python bench.py) > gpurun_out/r04/bench_r04.json 2> gpurun_out/r04/bench_r04.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_r04.json", errors="replace") if l.startswith("{")][-1])
print("headline", round(d["value"], 1), "frac", round(d["roofline"]["frac"], 4), "api", round(d["api"]["value"], 1), d["api"]["ms"])
print("decode", {k: d["decode"].get(k) for k in ("value", "ms", "every_byte_verified")}, d["decode"]["roofline"]["kernel"], round(d["decode"]["roofline"]["frac"], 4))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("bit_identical_vs_reference"), "cpu decode", d["decode"]["cpu_baseline"]["value"] if d["decode"].get("cpu_baseline") else None)
PY
export ZPAQ_AMD_SPLIT_COPY=0
(time timeout 200 python bench.py --cpu-seconds 0 --decode-blocks 0 --steps 1 --warmup 1 --verify-blocks 0) > gpurun_out/r04/bench_nosplit.json 2> gpurun_out/r04/bench_nosplit.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_nosplit.json", errors="replace") if l.startswith("{")][-1])
print("one copy up front: headline", round(d["value"], 1), "api", round(d["api"]["value"], 1), d["api"]["ms"])
PY
