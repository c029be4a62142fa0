#!/bin/bash
# GPU call 16 of round 6: call 15 found the latency shape of the LARGER chains slower (mid.cfg 108.7 -> 89.3 MB/s, -m5 on 64 / 256
# blocks 26.5 / 96.5): A/B of the coder with one store per bit against the windowed one on those chains, with unit profiles
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
L2="--legacy-level 2 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0"
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16"
timeout 300 python bench.py $L2 > $O/c16_legacy2_fast.json 2> $O/c16_legacy2_fast.err
ZPAQ_AMD_CODER_FAST=0 timeout 300 python bench.py $L2 > $O/c16_legacy2_window.json 2> $O/c16_legacy2_window.err
timeout 300 python bench.py $L2 > $O/c16_legacy2_fast_b.json 2> $O/c16_legacy2_fast_b.err
ZPAQ_AMD_PERSIST_PROF=$O/c16_prof_legacy2_fast.bin timeout 300 python bench.py $L2 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c16_prof_legacy2_fast.bin > $O/c16_prof_legacy2_fast.txt 2>&1
ZPAQ_AMD_CODER_FAST=0 ZPAQ_AMD_PERSIST_PROF=$O/c16_prof_legacy2_window.bin timeout 300 python bench.py $L2 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c16_prof_legacy2_window.bin > $O/c16_prof_legacy2_window.txt 2>&1
timeout 300 python bench.py $B --blocks 64 > $O/c16_m5_64_fast.json 2> $O/c16_m5_64_fast.err
ZPAQ_AMD_CODER_FAST=0 timeout 300 python bench.py $B --blocks 64 > $O/c16_m5_64_window.json 2> $O/c16_m5_64_window.err
timeout 300 python bench.py $B --blocks 256 > $O/c16_m5_256_fast.json 2> $O/c16_m5_256_fast.err
ZPAQ_AMD_CODER_FAST=0 timeout 300 python bench.py $B --blocks 256 > $O/c16_m5_256_window.json 2> $O/c16_m5_256_window.err
ZPAQ_AMD_PERSIST_PROF=$O/c16_prof_m5_64_fast.bin timeout 300 python bench.py $B --blocks 64 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c16_prof_m5_64_fast.bin > $O/c16_prof_m5_64_fast.txt 2>&1
ZPAQ_AMD_CODER_FAST=0 ZPAQ_AMD_PERSIST_PROF=$O/c16_prof_m5_64_window.bin timeout 300 python bench.py $B --blocks 64 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c16_prof_m5_64_window.bin > $O/c16_prof_m5_64_window.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c16_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -40 $O/c16_prof_legacy2_fast.txt
head -40 $O/c16_prof_legacy2_window.txt
