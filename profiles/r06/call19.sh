#!/bin/bash
# GPU call 19 of round 6: configs[1] three times on one box (call 17: 341 MB/s, call 18: 281 on another box with the ROW units
# 25 % slower) against the round's earlier form on the same box; mid.cfg as a small chain (threshold 32 wavefronts) against 16;
# the new GPU test of the small chains
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "small_chains or method_3 or legacy_models_at" > $O/c19_tests.txt 2>&1
tail -5 $O/c19_tests.txt
C1="--method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0"
timeout 300 python bench.py $C1 > $O/c19_configs1_a.json 2> $O/c19_configs1_a.err
ZPAQ_AMD_SMALL_CHAIN=0 ZPAQ_AMD_CODER_FAST=0 timeout 300 python bench.py $C1 --api-blocks 0 > $O/c19_configs1_old.json 2> $O/c19_configs1_old.err
timeout 300 python bench.py $C1 > $O/c19_configs1_b.json 2> $O/c19_configs1_b.err
timeout 300 python bench.py $C1 --api-blocks 0 > $O/c19_configs1_c.json 2> $O/c19_configs1_c.err
L2="--legacy-level 2 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0"
timeout 300 python bench.py $L2 > $O/c19_legacy2_16.json 2> $O/c19_legacy2_16.err
ZPAQ_AMD_SMALL_CHAIN_WAVES=32 timeout 300 python bench.py $L2 > $O/c19_legacy2_32.json 2> $O/c19_legacy2_32.err
timeout 300 python bench.py $L2 > $O/c19_legacy2_16_b.json 2> $O/c19_legacy2_16_b.err
ZPAQ_AMD_SMALL_CHAIN_WAVES=32 timeout 300 python bench.py $L2 > $O/c19_legacy2_32_b.json 2> $O/c19_legacy2_32_b.err
ZPAQ_AMD_SMALL_CHAIN_WAVES=32 ZPAQ_AMD_PERSIST_PROF=$O/c19_prof_legacy2_32.bin timeout 300 python bench.py $L2 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c19_prof_legacy2_32.bin > $O/c19_prof_legacy2_32.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c19_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), "api", (j.get("api") or {}).get("value"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -36 $O/c19_prof_legacy2_32.txt
