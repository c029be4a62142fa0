#!/bin/bash
# GPU call 22 of round 4: where the row wavefronts' nibble-start bits go (profile build with a forced vmcnt(0) between the stages).
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
export ZPAQ_AMD_SPEC_DEFS="-DZPQ_PROF -DZPQ_PROF2"
(time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_prof2.json 2> gpurun_out/r04/dec_prof2.err
grep -a "team prof" gpurun_out/r04/dec_prof2.json gpurun_out/r04/dec_prof2.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/dec_prof2.json", errors="replace") if l.startswith("{")][-1])
print("code ms", round(d["kernel_ms"]["code"], 1), "ok", d["all_status_ok"])
PY
