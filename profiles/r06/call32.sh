#!/bin/bash
# GPU call 32 of round 6: variant 3 with the small chains' LDS-rich ICM / ISSE maps (and the ROW units left with a lane per block)
# against variant 3 without (ZPAQ_AMD_WIDE_RICH=0): -m5 on 64 / 128 / 256 / 288 blocks, max.cfg on 64
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c32_$name.json 2> $O/c32_$name.err; }
for n in 64 128 256 288; do
  B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks $n"
  run m5_${n}_rich X=1
  run m5_${n}_plain ZPAQ_AMD_WIDE_RICH=0
done
B="--legacy-level 3 --kind text --blocks 64 --block-bytes 1048576 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0"
run legacy3_64_rich X=1
run legacy3_64_plain ZPAQ_AMD_WIDE_RICH=0
ZPAQ_AMD_PERSIST_PROF=$O/c32_prof_m5_64_rich.bin timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 64 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c32_prof_m5_64_rich.bin > $O/c32_prof_m5_64_rich.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c32_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
grep -E "^ +[0-9]+ +[0-9]+ +[0-9]+ +[0-9]+ " $O/c32_prof_m5_64_rich.txt | sort -k10 -n -r | awk '{print $5,$6,$7,$8,$9,$10,$12,$13}' | head -14
