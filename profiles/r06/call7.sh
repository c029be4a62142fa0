#!/bin/bash
# GPU call 7 of round 6: per-unit profiles of the headline under six packings (for the fit of the packing's cost model:
# a workgroup's time against what it holds)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 0 --warmup 0"
prof() { name=$1; shift; env "$@" ZPAQ_AMD_PERSIST_PROF=$O/c7_prof_$name.bin timeout 300 python bench.py $B > $O/c7_$name.json 2> $O/c7_$name.err; python profiles/persist_prof.py $O/c7_prof_$name.bin > $O/c7_prof_$name.txt 2>&1; }
prof def A=1
prof s0 ZPAQ_AMD_PACK_STREAM_WEIGHT=0
prof o03 ZPAQ_AMD_PACK_ONDIE_WEIGHT=0.3
prof o07 ZPAQ_AMD_PACK_ONDIE_WEIGHT=0.7
prof all1 ZPAQ_AMD_PACK_STREAM_WEIGHT=1 ZPAQ_AMD_PACK_ONDIE_WEIGHT=1
prof tables ZPAQ_AMD_PACK_LINES=0
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c7_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "code ms", round(j["kernel_ms"]["code"], 1), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e)
PY
