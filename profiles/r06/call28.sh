#!/bin/bash
# GPU call 28 of round 6: the lookahead that lost on the full machine (call 18), on the -m5 chains' LATENCY shape (64 / 256 / 512
# blocks: the machine is not full there): ICM / ISSE streams four bytes ahead, ROW table two bytes ahead, both, against the default
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c28_$name.json 2> $O/c28_$name.err; }
for n in 64 256 512; do
  B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks $n"
  run m5_${n}_def X=1
  run m5_${n}_ahead ZPAQ_AMD_STREAM_AHEAD_BIG=3
  run m5_${n}_ring ZPAQ_AMD_ROW_RING=1
  run m5_${n}_both ZPAQ_AMD_STREAM_AHEAD_BIG=3 ZPAQ_AMD_ROW_RING=1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c28_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
