#!/bin/bash
# GPU call 10 of round 6: (a) the API leg with the engine waiting for its own hashing kernel before the persistent launch (call 9:
# the launch stepped aside for sha1_blocks_kernel, api 267 MB/s); (b) who shares a SIMD inside a workgroup: k-th heaviest with the
# k-th lightest (default) against k-th with (k + 4)-th, alternating
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hashing_kernel or legacy_models_at_baseline" -s > $O/c10_tests.txt 2>&1
tail -4 $O/c10_tests.txt
timeout 600 python bench.py --cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --verify-blocks 64 > $O/c10_api.json 2> $O/c10_api.err
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64 --steps 2"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c10_$name.json 2> $O/c10_$name.err; }
for rep in a b c; do
run pair0_$rep A=1
run pair1_$rep ZPAQ_AMD_PACK_PAIRING=1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c10_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "code ms", round(j["kernel_ms"]["code"], 1), (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20],
              "api", {k: (j.get("api") or {}).get(k) for k in ("value", "persistent_launch", "persistent_launch_given_up_after_ms")})
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
