#!/bin/bash
# GPU call 3 of round 4: where the lockstep decoder's bit time goes (-DZPQ_PROF) and the touches two bits ahead, A/B.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
for v in default "-DZPQ_PROF"; do
  if [ "$v" = default ]; then unset ZPAQ_AMD_SPEC_DEFS; else export ZPAQ_AMD_SPEC_DEFS="$v"; fi
  n=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  (time timeout 240 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 262144 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0 --decode-blocks 0) > gpurun_out/r04/team_$n.json 2> gpurun_out/r04/team_$n.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r04/team_$n.json", errors="replace") if l.startswith("{")][-1])
print("$v", "MB/s", round(d["value"], 1), "code ms", round(d["kernel_ms"]["code"], 1), "ok", d["all_status_ok"], d["roofline"]["kernel_origin"])
PY
  grep -a "team prof" gpurun_out/r04/team_$n.err gpurun_out/r04/team_$n.json
done
