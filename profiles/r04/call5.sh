#!/bin/bash
# GPU call 5 of round 4: touches in the small-batch (latency-bound) regime: 64 and 256 blocks of 1 MiB, -m5.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
for nb in 64 256; do
for v in default "-DZPQ_PIPE_TOUCH=6"; do
  if [ "$v" = default ]; then unset ZPAQ_AMD_SPEC_DEFS; else export ZPAQ_AMD_SPEC_DEFS="$v"; fi
  n=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  (time timeout 200 python bench.py --blocks $nb --cpu-seconds 0 --api-blocks 0 --decode-blocks 0 --steps 1 --warmup 1 --verify-blocks 2) > gpurun_out/r04/enc${nb}_$n.json 2> gpurun_out/r04/enc${nb}_$n.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r04/enc${nb}_$n.json", errors="replace") if l.startswith("{")][-1])
print($nb, "$v", "MB/s", round(d["value"], 1), "code ms", round(d["kernel_ms"]["code"], 1), "ok", d["all_status_ok"], d["roundtrip_verified_blocks"], d["roofline"]["kernel_origin"])
PY
done
done
