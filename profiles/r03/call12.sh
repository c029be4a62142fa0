#!/bin/bash
# Round 3, GPU call 12 (last): the bench lines with the shipped code (roofline.traffic / transactions populated), the dense
# and the in-library lines, and the tests touched since the full run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c12
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt; }
line() {
  python - <<PY | tee -a $O/summary.txt
import json
try:
    j = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    a = j.get("api") or {}; c = j.get("cpu_baseline") or {}; r = j["roofline"]
    print("%-10s n_gpus=%s value=%7.1f MB/s frac=%.4f traffic=%s transactions=%s ok=%s api=%s cpu=%s" % ("$1", j["n_gpus"], j["value"], r["frac"], r.get("traffic"),
          {k: round(v, 2) for k, v in (r.get("transactions") or {}).items() if isinstance(v, float)}, j["all_status_ok"], a.get("value"), c.get("value")))
except Exception as e:
    print("$1 FAILED", e, open("$O/bench_$1.err").read()[-400:])
PY
}
echo "== bench lines" | tee $O/summary.txt
timeout 500 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err; line headline; stamp
timeout 300 python bench.py --blocks 2048 --cpu-seconds 0 --api-blocks 0 > $O/bench_dense.json 2> $O/bench_dense.err; line dense; stamp
ZPAQ_AMD_DEVICES=0,0 timeout 400 python bench.py --gpus 2 --in-library --blocks 256 --cpu-seconds 0 > $O/bench_inlib.json 2> $O/bench_inlib.err; line inlib; stamp
echo "== tests" | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "both_shapes or suffix_arrays or state_budget or two_engines or large_batch" > $O/tests.txt 2>&1
tail -3 $O/tests.txt | tee -a $O/summary.txt; stamp
