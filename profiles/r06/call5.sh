#!/bin/bash
# GPU call 5 of round 6: the packing of a group's unit wavefronts into workgroups balanced by MEMORY LINES per input byte (call 2's
# per-unit profile: a workgroup's wavefronts are all as slow as the workgroup's line count is high -- 288 lines: 2.5-2.8 s,
# 160: 1.6-1.9 s) against the table-first packing; lines that stay on the die at half and at full weight
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c5_$name.json 2> $O/c5_$name.err; }
run lines_w05 A=1
run lines_w10 ZPAQ_AMD_PACK_ONDIE_WEIGHT=1
run tables_first ZPAQ_AMD_PACK_LINES=0
run lines_w05_again A=1
run lines_packed_lds128 ZPAQ_AMD_MIX_PACKED=1
run lines_packed_lds0 ZPAQ_AMD_MIX_PACKED=1 ZPAQ_AMD_MIX_LDS_ROWS=0
ZPAQ_AMD_PERSIST_PROF=$O/c5_prof_lines_w05.bin timeout 300 python bench.py $B --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c5_prof_lines_w05.bin > $O/c5_prof_lines_w05.txt 2>&1
ZPAQ_AMD_PACK_ONDIE_WEIGHT=1 ZPAQ_AMD_PERSIST_PROF=$O/c5_prof_lines_w10.bin timeout 300 python bench.py $B --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c5_prof_lines_w10.bin > $O/c5_prof_lines_w10.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c5_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
grep -A9 "per workgroup flavour" $O/c5_prof_lines_w05.txt
grep -A9 "per workgroup flavour" $O/c5_prof_lines_w10.txt
cat $O/c5_prof_lines_w05.txt | head -70
