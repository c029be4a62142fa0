#!/bin/bash
# GPU call 21 of round 4: lockstep decoder with the row wavefronts' global stores behind [A] (LATE_STORES) on top of call 20's
# best (12 871.7 ms); phase profile of the same build.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
for v in 1 0; do
  export ZPAQ_AMD_SPEC_DEFS="-DZPQ_TEAM_LATE_STORES=$v"
  (time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_ls_$v.json 2> gpurun_out/r04/dec_ls_$v.err
  python - $v <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/r04/dec_ls_%s.json" % sys.argv[1], errors="replace") if l.startswith("{")][-1])
print("late stores =", sys.argv[1], "code ms", round(d["kernel_ms"]["code"], 1), "MB/s", round(2048 * 1.048576 / (d["kernel_ms"]["code"] / 1e3), 1), "ok", d["all_status_ok"], d["roofline"]["kernel_origin"][:60])
PY
done
export ZPAQ_AMD_SPEC_DEFS=-DZPQ_PROF
(time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_ls_prof.json 2> gpurun_out/r04/dec_ls_prof.err
grep -a "team prof" gpurun_out/r04/dec_ls_prof.json gpurun_out/r04/dec_ls_prof.err
