#!/bin/bash
# GPU call 18 of round 4: the lockstep decoder with packed side tables (14 of 16 in LDS instead of 7) and the compact stretch table: the configs[4] point and the phase profile.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
(time timeout 300 python bench.py --mode decode --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_packed.json 2> gpurun_out/r04/dec_packed.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/dec_packed.json", errors="replace") if l.startswith("{")][-1])
print("decode 2048 x 1 MiB", "code ms", round(d["kernel_ms"]["code"], 1), "MB/s kernel", round(2048 * 1.048576 / (d["kernel_ms"]["code"] / 1e3), 1), "ok", d["all_status_ok"], d["roofline"]["kernel_origin"])
PY
export ZPAQ_AMD_SPEC_DEFS=-DZPQ_PROF
(time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_packed_prof.json 2> gpurun_out/r04/dec_packed_prof.err
grep -a "team prof" gpurun_out/r04/dec_packed_prof.json gpurun_out/r04/dec_packed_prof.err
