#!/bin/bash
# GPU call 20 of round 4: lockstep decoder A/B -- the second nibble's rows loaded (not touched) one bit early (EARLY2), the
# mixers' update of a byte's last bit behind HCOMP and [C] (LATE_UPDATE7); both builds also carry the branch-free
# decode_bit / ISSE second-word store.  Code objects prebuilt by profiles/r04/prebuild_variants.py.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
for v in "00" "10" "01" "11"; do
  export ZPAQ_AMD_SPEC_DEFS="-DZPQ_TEAM_EARLY2=${v:0:1} -DZPQ_TEAM_LATE_UPDATE7=${v:1:1}"
  (time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_ab_$v.json 2> gpurun_out/r04/dec_ab_$v.err
  python - $v <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/r04/dec_ab_%s.json" % sys.argv[1], errors="replace") if l.startswith("{")][-1])
print("early2/late7 =", sys.argv[1], "code ms", round(d["kernel_ms"]["code"], 1), "MB/s", round(2048 * 1.048576 / (d["kernel_ms"]["code"] / 1e3), 1), "ok", d["all_status_ok"], d["roofline"]["kernel_origin"][:60])
PY
done
export ZPAQ_AMD_SPEC_DEFS=-DZPQ_PROF
(time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_ab_prof.json 2> gpurun_out/r04/dec_ab_prof.err
grep -a "team prof" gpurun_out/r04/dec_ab_prof.json gpurun_out/r04/dec_ab_prof.err
