#!/bin/bash
# Round 3, GPU call 4: where do the latency-optimised units (lanes per bit / nibble) beat the lane-per-block units?  -m5, 1 MiB text blocks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c4
mkdir -p $O
cd $R
for nb in 64 128 256 512 2048; do
  echo "== $nb blocks" | tee -a $O/summary.txt
  timeout 400 python profiles/ab_inproc.py profiles/r03/ab3_small_$nb.json --out $O/ab3_$nb.jsonl > $O/ab3_$nb.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for ln in open("$O/ab3_$nb.jsonl"):
    j = json.loads(ln)
    if "error" in j: print("%-18s ERROR %s" % (j["name"], j["error"][:160])); continue
    print("%-18s %8.1f ms %7.1f MB/s ok=%s same=%s" % (j["name"], j["code_ms"], j["MBps"], j["status_ok"], j["same_bytes_as_first"]))
PY
done
