#!/bin/bash
# GPU call 4 of round 4: the encoder's lane-per-block units with touches T bytes ahead (ROW, MIX 16, MATCH index), A/B on the headline.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
for v in default "-DZPQ_PIPE_TOUCH=3" "-DZPQ_PIPE_TOUCH=6"; do
  if [ "$v" = default ]; then unset ZPAQ_AMD_SPEC_DEFS; else export ZPAQ_AMD_SPEC_DEFS="$v"; fi
  n=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  (time timeout 200 python bench.py --cpu-seconds 0 --api-blocks 0 --decode-blocks 0 --steps 2 --warmup 1 --verify-blocks 2) > gpurun_out/r04/enc_$n.json 2> gpurun_out/r04/enc_$n.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r04/enc_$n.json", errors="replace") if l.startswith("{")][-1])
print("$v", "MB/s", round(d["value"], 1), "code ms", round(d["kernel_ms"]["code"], 1), "ok", d["all_status_ok"], d["roundtrip_verified_blocks"], d["roofline"]["kernel_origin"], "ratio", d["ratio"])
PY
done
