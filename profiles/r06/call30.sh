#!/bin/bash
# GPU call 30 of round 6: the -m5 chains' latency shape with a wavefront per SIMD AND the small chains' LDS-rich units
# (ZPAQ_AMD_SMALL_CHAIN_WAVES=400 ZPAQ_AMD_SMALL_CHAIN_W4_WAVES=400) against the wavefront per SIMD alone (ZPAQ_AMD_LATENCY_W4=1)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c30_$name.json 2> $O/c30_$name.err; }
for n in 64 128 256; do
  B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks $n"
  run m5_${n}_w4 ZPAQ_AMD_LATENCY_W4=1
  run m5_${n}_richw4 ZPAQ_AMD_SMALL_CHAIN_WAVES=400 ZPAQ_AMD_SMALL_CHAIN_W4_WAVES=400
done
ZPAQ_AMD_SMALL_CHAIN_WAVES=400 ZPAQ_AMD_SMALL_CHAIN_W4_WAVES=400 ZPAQ_AMD_PERSIST_PROF=$O/c30_prof_m5_64_richw4.bin timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 64 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c30_prof_m5_64_richw4.bin > $O/c30_prof_m5_64_richw4.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c30_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -3 $O/c30_prof_m5_64_richw4.txt
grep -E "^ +[0-9]+ +[0-9]+ +[0-9]+ +[0-9]+ " $O/c30_prof_m5_64_richw4.txt | sort -k10 -n -r | head -12
