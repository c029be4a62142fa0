#!/bin/bash
# GPU call 6 of round 6: the line-balanced packing with the streams' lines in the model (weights: lines that stay on the die,
# stream lines), each variant twice in alternation (run-to-run spread on one box: ~2 %)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64 --steps 2"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c6_$name.json 2> $O/c6_$name.err; }
for rep in a b; do
run def_$rep A=1
run s0_$rep ZPAQ_AMD_PACK_STREAM_WEIGHT=0
run s05_$rep ZPAQ_AMD_PACK_STREAM_WEIGHT=0.5
run o03_$rep ZPAQ_AMD_PACK_ONDIE_WEIGHT=0.3
run o07_$rep ZPAQ_AMD_PACK_ONDIE_WEIGHT=0.7
run all1_$rep ZPAQ_AMD_PACK_STREAM_WEIGHT=1 ZPAQ_AMD_PACK_ONDIE_WEIGHT=1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c6_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "code ms", round(j["kernel_ms"]["code"], 1), (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
