#!/bin/bash
# GPU call 8 of round 6: the packing by FITTED loads (profiles/r06/fit_packing.py) against the table-first packing, alternating
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64 --steps 2"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c8_$name.json 2> $O/c8_$name.err; }
for rep in a b c; do
run fitted_$rep A=1
run tables_$rep ZPAQ_AMD_PACK_LINES=0
done
ZPAQ_AMD_PERSIST_PROF=$O/c8_prof_fitted.bin timeout 300 python bench.py $B --warmup 0 --steps 1 > /dev/null 2>&1
python profiles/persist_prof.py $O/c8_prof_fitted.bin > $O/c8_prof_fitted.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c8_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "code ms", round(j["kernel_ms"]["code"], 1), (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
grep -A9 "per workgroup flavour" $O/c8_prof_fitted.txt
