#!/bin/bash
# GPU call 13 of round 4: the other BASELINE configurations on one GPU with the CPU reference beside each (round 3's copies of
# these lines had cpu_baseline null): configs[1] (-m3, 256 x 256 KiB LCG), configs[3]'s corpus (mixed, 1024 x 1 MiB), the
# 2048-block encode.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
run() {
  n=$1; shift
  (time timeout 400 python bench.py "$@") > gpurun_out/r04/bench_$n.json 2> gpurun_out/r04/bench_$n.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_$n.json", errors="replace") if l.startswith("{")][-1])
a = d.get("api") or {}
c = d.get("cpu_baseline") or {}
print("$n", "MB/s", round(d["value"], 1), "frac", round(d["roofline"]["frac"], 4), "api", a.get("value"), "cpu", c.get("value"), c.get("bit_identical_vs_reference"), "ok", d["all_status_ok"], d["config"]["workload"][:90])
PY
}
run configs1 --method 3 --blocks 256 --block-bytes 262144 --kind lcg --decode-blocks 0 --steps 3
run mixed --kind mixed --decode-blocks 0

