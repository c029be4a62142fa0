#!/bin/bash
# GPU call 24 of round 4: lockstep decoder A/B -- HCOMP's M array behind a register window (MWIN; re-measured now that the last
# bit's update no longer runs in front of HCOMP), one lane per block stores into H (HWRITE1).
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
for v in "11" "00" "10" "01"; do
  export ZPAQ_AMD_SPEC_DEFS="-DZPQ_TEAM_MWIN=${v:0:1} -DZPQ_TEAM_HWRITE1=${v:1:1}"
  if [ $v = 11 ]; then unset ZPAQ_AMD_SPEC_DEFS; fi
  (time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_mw_$v.json 2> gpurun_out/r04/dec_mw_$v.err
  python - $v <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/r04/dec_mw_%s.json" % sys.argv[1], errors="replace") if l.startswith("{")][-1])
print("mwin/hwrite1 =", sys.argv[1], "code ms", round(d["kernel_ms"]["code"], 1), "MB/s", round(2048 * 1.048576 / (d["kernel_ms"]["code"] / 1e3), 1), "ok", d["all_status_ok"], d["roofline"]["kernel_origin"][:60])
PY
done
export ZPAQ_AMD_SPEC_DEFS=-DZPQ_PROF
(time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_mw_prof.json 2> gpurun_out/r04/dec_mw_prof.err
grep -a "team prof" gpurun_out/r04/dec_mw_prof.json gpurun_out/r04/dec_mw_prof.err
