#!/bin/bash
# GPU call 3 of round 6: the MIX units in halves (call 2: +2 % on the round-5 units: the launch is no longer paced by ONE unit)
# with packed rows -- 64-byte rows for m16, rows 0 .. 127 / 0 .. 63 of m8 in LDS -- against unpacked halves and round 5's form;
# configs[1] with the coder's shift-out loop in closed form
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c3_$name.json 2> $O/c3_$name.err; }
run halves_lds128 ZPAQ_AMD_MIX_HALVES=1
run halves_lds64 ZPAQ_AMD_MIX_HALVES=1 ZPAQ_AMD_MIX_LDS_ROWS=64
run halves_lds0 ZPAQ_AMD_MIX_HALVES=1 ZPAQ_AMD_MIX_LDS_ROWS=0
run halves_unpacked ZPAQ_AMD_MIX_HALVES=1 ZPAQ_AMD_MIX_PACKED=0
run unpacked ZPAQ_AMD_MIX_PACKED=0
ZPAQ_AMD_MIX_HALVES=1 ZPAQ_AMD_PERSIST_PROF=$O/c3_prof_halves_lds128.bin timeout 300 python bench.py $B --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c3_prof_halves_lds128.bin > $O/c3_prof_halves_lds128.txt 2>&1
timeout 300 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 3 > $O/c3_configs1.json 2> $O/c3_configs1.err
ZPAQ_AMD_PERSIST_PROF=$O/c3_prof_configs1.bin timeout 300 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c3_prof_configs1.bin > $O/c3_prof_configs1.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c3_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), (j.get("api") or {}).get("value"), (j.get("cpu_baseline") or {}).get("bit_identical_vs_reference"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
cat $O/c3_prof_configs1.txt | head -20
cat $O/c3_prof_halves_lds128.txt | head -75
