#!/bin/bash
# GPU call 33 of round 6: the final code (variant 3 with the LDS-rich ICM / ISSE maps): the whole GPU suite, the default line
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/c33_gputest.txt 2>&1
tail -10 $O/c33_gputest.txt
timeout 1500 python bench.py > $O/c33_bench_default.json 2> $O/c33_bench_default.err
python - <<PY
import json
j = json.loads([l for l in open("$O/c33_bench_default.json") if l.startswith("{")][-1])
print("default", round(j["value"], 1), "frac", round(j["roofline"]["frac"], 4), "traffic", j["roofline"]["traffic"], "origin", j["roofline"]["kernel_origin"][:20], "api", (j.get("api") or {}).get("value"), "decode", (j.get("decode") or {}).get("value"),
      "configs1", (j.get("configs1") or {}).get("value"), "legacy2", (j.get("legacy2") or {}).get("value"), "legacy3", (j.get("legacy3") or {}).get("value"), "ident", (j.get("reference_identity") or {}).get("identical"))
PY
