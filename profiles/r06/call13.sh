#!/bin/bash
# GPU call 13 of round 6: m8's rows 0 .. 63 in LDS in their 32-bit form (no packing arithmetic), halves, line-balanced packing
# (ZPAQ_AMD_MIX_LDS32 existed only in the working tree of this call: the generalisation of pipe_mix_packed_unit to 32-bit quads; removed after the measurement)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --steps 2"
run() { name=$1; shift; env "$@" timeout 600 python bench.py $B > $O/c13_$name.json 2> $O/c13_$name.err; }
for rep in a b c; do
run def_$rep A=1
run lds32_64_$rep ZPAQ_AMD_MIX_LDS32=1
run lds32_32_$rep ZPAQ_AMD_MIX_LDS32=1 ZPAQ_AMD_MIX_LDS_ROWS=32
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c13_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1), (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:22])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
