#!/bin/bash
# GPU call 29 of round 6: a wavefront per SIMD (workgroups of 4) for the -m5 chains' latency shape, without the small chains'
# LDS-rich units (ZPAQ_AMD_LATENCY_W4=1), on 64 / 128 / 256 blocks; and where the two shapes cross: 384 blocks in either
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c29_$name.json 2> $O/c29_$name.err; }
for n in 64 128 256; do
  B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks $n"
  run m5_${n}_def X=1
  run m5_${n}_w4 ZPAQ_AMD_LATENCY_W4=1
done
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 384"
run m5_384_latency ZPAQ_AMD_PIPE_MODE=latency
run m5_384_throughput ZPAQ_AMD_PIPE_MODE=throughput
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 288"
run m5_288_latency ZPAQ_AMD_PIPE_MODE=latency
run m5_288_throughput ZPAQ_AMD_PIPE_MODE=throughput
ZPAQ_AMD_LATENCY_W4=1 ZPAQ_AMD_PERSIST_PROF=$O/c29_prof_m5_64_w4.bin timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 64 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c29_prof_m5_64_w4.bin > $O/c29_prof_m5_64_w4.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c29_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
grep -E "^ +[0-9]+ +[0-9]+ +[0-9]+ +[0-9]+ " $O/c29_prof_m5_64_w4.txt | sort -k10 -n -r | head -12
