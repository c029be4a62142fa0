#!/bin/bash
# GPU call 18 of round 6: HCOMP with M in LDS and the input a word ahead (configs[1], mid.cfg with profiles); then the HEADLINE
# with what call 17 taught -- in-order vmcnt behind write-through stores: the ICM / ISSE maps' streams four bytes ahead
# (ZPAQ_AMD_STREAM_AHEAD_BIG=3), the ROW units' table two bytes ahead (ZPAQ_AMD_ROW_RING=1), both, alternating with the default
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
C1="--method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0"
timeout 300 python bench.py $C1 --cpu-seconds 3 > $O/c18_configs1.json 2> $O/c18_configs1.err
ZPAQ_AMD_PERSIST_PROF=$O/c18_prof_configs1.bin timeout 300 python bench.py $C1 --cpu-seconds 0 --api-blocks 0 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c18_prof_configs1.bin > $O/c18_prof_configs1.txt 2>&1
L2="--legacy-level 2 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0"
timeout 300 python bench.py $L2 > $O/c18_legacy2.json 2> $O/c18_legacy2.err
ZPAQ_AMD_PERSIST_PROF=$O/c18_prof_legacy2.bin timeout 300 python bench.py $L2 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c18_prof_legacy2.bin > $O/c18_prof_legacy2.txt 2>&1
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c18_$name.json 2> $O/c18_$name.err; }
run head_def_a X=1
run head_ahead_a ZPAQ_AMD_STREAM_AHEAD_BIG=3
run head_ring_a ZPAQ_AMD_ROW_RING=1
run head_both_a ZPAQ_AMD_STREAM_AHEAD_BIG=3 ZPAQ_AMD_ROW_RING=1
run head_def_b X=1
run head_both_b ZPAQ_AMD_STREAM_AHEAD_BIG=3 ZPAQ_AMD_ROW_RING=1
run head_ring_b ZPAQ_AMD_ROW_RING=1
ZPAQ_AMD_STREAM_AHEAD_BIG=3 ZPAQ_AMD_ROW_RING=1 ZPAQ_AMD_PERSIST_PROF=$O/c18_prof_head_both.bin timeout 300 python bench.py $B --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c18_prof_head_both.bin > $O/c18_prof_head_both.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c18_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), "api", (j.get("api") or {}).get("value"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -12 $O/c18_prof_configs1.txt
head -30 $O/c18_prof_legacy2.txt
