#!/bin/bash
# GPU call 12 of round 6: MATCH's share in the packing's line count (2 lines per block and byte; the fit says a MATCH wavefront
# weighs 1.36 ROW wavefronts, and the sweep's 4 MiB zeros line -- MATCH's worst case -- went down with the line-balanced packing)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --steps 2"
Z="--kind zeros --block-bytes 4194304 --cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 4 --verify-bytes 65536 --steps 1"
run() { name=$1; shift; env "$@" timeout 600 python bench.py $B > $O/c12_text_$name.json 2> $O/c12_text_$name.err; }
runz() { name=$1; shift; env "$@" timeout 600 python bench.py $Z > $O/c12_zeros_$name.json 2> $O/c12_zeros_$name.err; }
for rep in a b; do
run m2_$rep A=1
run m3_$rep ZPAQ_AMD_PACK_MATCH_LINES=3
run m4_$rep ZPAQ_AMD_PACK_MATCH_LINES=4
done
runz m2 A=1
runz m3 ZPAQ_AMD_PACK_MATCH_LINES=3
runz m4 ZPAQ_AMD_PACK_MATCH_LINES=4
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c12_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1), j["roofline"]["kernel_origin"][:22])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
