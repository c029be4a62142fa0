#!/bin/bash
# GPU call 35 of round 6: the final code (every latency-shape chain with the LDS-rich ICM / ISSE maps): small-batch ladder of -m5,
# the whole GPU suite, the default line
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c35_$name.json 2> $O/c35_$name.err; }
for n in 64 128 256 288 320 384 512 576 640; do
  B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks $n"
  run m5_${n} X=1
done
B="--kind mixed --blocks 256 --cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16"
run mixed_256 X=1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c35_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/c35_gputest.txt 2>&1
tail -10 $O/c35_gputest.txt
timeout 1500 python bench.py > $O/c35_bench_default.json 2> $O/c35_bench_default.err
python - <<PY
import json
j = json.loads([l for l in open("$O/c35_bench_default.json") if l.startswith("{")][-1])
print("default", round(j["value"], 1), "frac", round(j["roofline"]["frac"], 4), "traffic", j["roofline"]["traffic"], "origin", j["roofline"]["kernel_origin"][:20], "api", (j.get("api") or {}).get("value"), "decode", (j.get("decode") or {}).get("value"),
      "configs1", (j.get("configs1") or {}).get("value"), "legacy2", (j.get("legacy2") or {}).get("value"), "legacy3", (j.get("legacy3") or {}).get("value"), "ident", (j.get("reference_identity") or {}).get("identical"))
PY
