#!/bin/bash
# GPU call 27 of round 6: the default line once more (now that traffic.json has the final code object's counters) and the
# north-star sweep on the final code
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 1500 python bench.py > $O/c27_bench_default.json 2> $O/c27_bench_default.err
python - <<PY
import json
j = json.loads([l for l in open("$O/c27_bench_default.json") if l.startswith("{")][-1])
print("default", round(j["value"], 1), "frac", round(j["roofline"]["frac"], 4), "traffic", j["roofline"]["traffic"], "transactions", j["roofline"].get("transactions"), "api", (j.get("api") or {}).get("value"), "decode", (j.get("decode") or {}).get("value"),
      "configs1", (j.get("configs1") or {}).get("value"), "legacy2", (j.get("legacy2") or {}).get("value"), "legacy3", (j.get("legacy3") or {}).get("value"), "ident", (j.get("reference_identity") or {}).get("identical"))
PY
timeout 1500 python profiles/sweep_north.py $O/c27_sweep_north.jsonl > $O/c27_sweep.log 2>&1
python - <<PY
import json
for ln in open("$O/c27_sweep_north.jsonl"):
    j = json.loads(ln)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.items() if k in ("block_bytes", "blocks", "kind", "MBps", "roofline_frac", "cpu_MBps", "ok", "decoded_back", "blocks_identical_to_reference", "error", "skipped")})
PY
