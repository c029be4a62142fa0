#!/bin/bash
# GPU call 14 of round 4: the north-star sweep again, every block the reference codes compared with the device's payload.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
(time timeout 780 python profiles/sweep_north.py gpurun_out/r04/sweep_north.jsonl) > gpurun_out/r04/sweep.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r04/sweep_north.jsonl"):
    d = json.loads(l)
    print(d["block_bytes"], d["blocks"], d["kind"], round(d.get("MBps", 0), 1), round(d.get("roofline_frac", 0), 4), "cpu", round(d.get("cpu_MBps", 0), 1), "ok", d.get("ok"), "back", d.get("decoded_back"), "identical", d.get("blocks_identical_to_reference"), "/", d.get("blocks_compared_with_reference"), d.get("error", ""))
PY
