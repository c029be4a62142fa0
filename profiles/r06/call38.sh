#!/bin/bash
# GPU call 38 of round 6: what made call 34's form faster -- the ICM maps' whole stretch tables or the 16 workgroups per group?
# -m5 on 384 / 512 blocks: default (14 workgroups per group, compact stretch), 16 workgroups per group alone
# (ZPAQ_AMD_PERSIST_WPG_MIN=16), whole stretch tables (ZPAQ_AMD_LATENCY_ICM_FULL=1: 16 per group as well)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c38_$name.json 2> $O/c38_$name.err; }
for n in 384 512; do
  B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks $n"
  run m5_${n}_def X=1
  run m5_${n}_wpg16 ZPAQ_AMD_PERSIST_WPG_MIN=16
  run m5_${n}_icmfull ZPAQ_AMD_LATENCY_ICM_FULL=1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c38_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
