#!/bin/bash
# GPU call 17 of round 6: coder with a per-byte window (one store per input byte instead of one per bit), small chains with
# their streams read four bytes ahead in rings of statically indexed slots (in-order vmcnt behind write-through stores), ROW
# units two bytes ahead in the table, HCOMP's stream stores behind the program.  configs[1] with profile and A/B of the
# lookahead; mid.cfg and -m5 small batches with the new coder against the windowed one; encoder parity tests; the headline.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
C1="--method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0"
timeout 300 python bench.py $C1 --cpu-seconds 3 > $O/c17_configs1.json 2> $O/c17_configs1.err
ZPAQ_AMD_PERSIST_PROF=$O/c17_prof_configs1.bin timeout 300 python bench.py $C1 --cpu-seconds 0 --api-blocks 0 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c17_prof_configs1.bin > $O/c17_prof_configs1.txt 2>&1
ZPAQ_AMD_STREAM_AHEAD=0 timeout 300 python bench.py $C1 --cpu-seconds 0 --api-blocks 0 > $O/c17_configs1_ahead0.json 2> $O/c17_configs1_ahead0.err
ZPAQ_AMD_STREAM_AHEAD=0 ZPAQ_AMD_PERSIST_PROF=$O/c17_prof_configs1_ahead0.bin timeout 300 python bench.py $C1 --cpu-seconds 0 --api-blocks 0 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c17_prof_configs1_ahead0.bin > $O/c17_prof_configs1_ahead0.txt 2>&1
L2="--legacy-level 2 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0"
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16"
timeout 300 python bench.py $L2 > $O/c17_legacy2_fast.json 2> $O/c17_legacy2_fast.err
ZPAQ_AMD_CODER_FAST=0 timeout 300 python bench.py $L2 > $O/c17_legacy2_window.json 2> $O/c17_legacy2_window.err
ZPAQ_AMD_PERSIST_PROF=$O/c17_prof_legacy2_fast.bin timeout 300 python bench.py $L2 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c17_prof_legacy2_fast.bin > $O/c17_prof_legacy2_fast.txt 2>&1
timeout 300 python bench.py $B --blocks 64 > $O/c17_m5_64_fast.json 2> $O/c17_m5_64_fast.err
ZPAQ_AMD_CODER_FAST=0 timeout 300 python bench.py $B --blocks 64 > $O/c17_m5_64_window.json 2> $O/c17_m5_64_window.err
timeout 300 python bench.py $B --blocks 256 > $O/c17_m5_256_fast.json 2> $O/c17_m5_256_fast.err
ZPAQ_AMD_CODER_FAST=0 timeout 300 python bench.py $B --blocks 256 > $O/c17_m5_256_window.json 2> $O/c17_m5_256_window.err
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=5 -k "encode_matches or compress_blocks_bit or nine_component or legacy or both_shapes or method_3 or mixed_plans or zeros_known or persistent_launch_gives or device_resident or random_hcomp" > $O/c17_tests.txt 2>&1
tail -5 $O/c17_tests.txt
timeout 400 python bench.py $B --verify-blocks 64 > $O/c17_head_a.json 2> $O/c17_head_a.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c17_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), "api", (j.get("api") or {}).get("value"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -12 $O/c17_prof_configs1.txt
head -12 $O/c17_prof_configs1_ahead0.txt
head -30 $O/c17_prof_legacy2_fast.txt
