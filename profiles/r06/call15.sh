#!/bin/bash
# GPU call 15 of round 6 (second session): the latency shape's per-bit instruction streams cut -- coder with one store per bit
# (pipe_coder_fast), pipe_find without the switch ladders (every ROW unit, both shapes), small chains (<= 16 unit wavefronts:
# configs[1]'s n = 2) with a SIMD per wavefront, unpacked ISSE pairs and whole squash / stretch tables in LDS.
# Parity first (encoder tests), then configs[1] with its unit profile, the legacy mid model, small -m5 batches, the headline twice.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=5 -k "encode_matches or compress_blocks_bit or nine_component or legacy or both_shapes or method_3 or mixed_plans or zeros_known or persistent_launch_gives or device_resident or random_hcomp" > $O/c15_tests.txt 2>&1
tail -12 $O/c15_tests.txt
C1="--method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0"
timeout 300 python bench.py $C1 --cpu-seconds 3 > $O/c15_configs1.json 2> $O/c15_configs1.err
ZPAQ_AMD_PERSIST_PROF=$O/c15_prof_configs1.bin timeout 300 python bench.py $C1 --cpu-seconds 0 --api-blocks 0 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c15_prof_configs1.bin > $O/c15_prof_configs1.txt 2>&1
ZPAQ_AMD_CODER_FAST=0 ZPAQ_AMD_SMALL_CHAIN=0 timeout 300 python bench.py $C1 --cpu-seconds 0 --api-blocks 0 > $O/c15_configs1_old.json 2> $O/c15_configs1_old.err
ZPAQ_AMD_SMALL_CHAIN=0 timeout 300 python bench.py $C1 --cpu-seconds 0 --api-blocks 0 > $O/c15_configs1_fastcoder_only.json 2> $O/c15_configs1_fastcoder_only.err
timeout 300 python bench.py --legacy-level 2 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0 > $O/c15_legacy2.json 2> $O/c15_legacy2.err
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64"
timeout 300 python bench.py $B --blocks 64 > $O/c15_m5_64.json 2> $O/c15_m5_64.err
timeout 300 python bench.py $B --blocks 256 > $O/c15_m5_256.json 2> $O/c15_m5_256.err
timeout 400 python bench.py $B > $O/c15_head_a.json 2> $O/c15_head_a.err
timeout 400 python bench.py $B > $O/c15_head_b.json 2> $O/c15_head_b.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c15_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), "api", (j.get("api") or {}).get("value"), (j.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -20 $O/c15_prof_configs1.txt
