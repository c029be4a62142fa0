#!/bin/bash
# GPU call 37 of round 6: variant 1 of the larger chains with the LDS-rich ISSE maps but the compact stretch in its ICM maps (14
# workgroups per group again): the -m5 ladder 320 .. 640 blocks, the archiver (two chains side by side), then the whole GPU suite
# and the default line on what is now the final code
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c37_$name.json 2> $O/c37_$name.err; }
for n in 64 256 320 384 512 576 640; do
  B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks $n"
  run m5_${n} X=1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c37_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
timeout 900 python profiles/r05/cli_bench.py --files 256 --out $O/c37_cli.json > $O/c37_cli.log 2>&1
cut -c1-200 $O/c37_cli.log | tail -8
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/c37_gputest.txt 2>&1
tail -10 $O/c37_gputest.txt
timeout 1500 python bench.py > $O/c37_bench_default.json 2> $O/c37_bench_default.err
python - <<PY
import json
j = json.loads([l for l in open("$O/c37_bench_default.json") if l.startswith("{")][-1])
print("default", round(j["value"], 1), "frac", round(j["roofline"]["frac"], 4), "traffic", j["roofline"]["traffic"], "origin", j["roofline"]["kernel_origin"][:20], "api", (j.get("api") or {}).get("value"), "decode", (j.get("decode") or {}).get("value"),
      "configs1", (j.get("configs1") or {}).get("value"), "legacy2", (j.get("legacy2") or {}).get("value"), "legacy3", (j.get("legacy3") or {}).get("value"), "ident", (j.get("reference_identity") or {}).get("identical"))
PY
