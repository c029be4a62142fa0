#!/bin/bash
# Round 3, GPU call 5: padded MIX rows (one 128-byte line per row); small batches with 8 hardware queues
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c5
mkdir -p $O
cd $R
for q in 4 8; do
for nb in 64 256 1024; do
  echo "== $nb blocks, GPU_MAX_HW_QUEUES=$q" | tee -a $O/summary.txt
  GPU_MAX_HW_QUEUES=$q timeout 400 python profiles/ab_inproc.py profiles/r03/ab5_$nb.json --out $O/ab5_${nb}_q$q.jsonl > $O/ab5_${nb}_q$q.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for ln in open("$O/ab5_${nb}_q$q.jsonl"):
    j = json.loads(ln)
    if "error" in j: print("%-18s ERROR %s" % (j["name"], j["error"][:160])); continue
    print("%-18s %8.1f ms %7.1f MB/s ok=%s same=%s" % (j["name"], j["code_ms"], j["MBps"], j["status_ok"], j["same_bytes_as_first"]))
PY
done
done
echo "== configs[1]" | tee -a $O/summary.txt
BENCH="python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --cpu-seconds 0 --steps 3 --warmup 1"
run() {
  local name=$1; shift
  env "$@" timeout 300 $BENCH > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("%-22s value=%7.1f MB/s code_ms=%8.1f ok=%s api=%s" % ("$name", j["value"], j["kernel_ms"]["code"], j["all_status_ok"], (j.get("api") or {}).get("value")))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-300:])
PY
}
run c1_q8_default GPU_MAX_HW_QUEUES=8
run c1_q8_rows_d3 GPU_MAX_HW_QUEUES=8 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=3
run c1_q8_rows_d2 GPU_MAX_HW_QUEUES=8 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=2
