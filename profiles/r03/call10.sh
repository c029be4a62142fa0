#!/bin/bash
# Round 3, GPU call 10: the latency shape with 2048-byte steps (blocks of 128 KiB and more) against call 6's numbers
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c10
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt; }
echo "== parity of both shapes (long blocks take the 2048-byte steps)" | tee $O/summary.txt
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "both_shapes or method_3_known or 4_mib_zeros" > $O/parity.txt 2>&1
tail -3 $O/parity.txt | tee -a $O/summary.txt; stamp
for nb in 64 256 512 640; do
  echo "== $nb blocks" | tee -a $O/summary.txt
  timeout 300 python profiles/ab_inproc.py profiles/r03/ab10_$nb.json --out $O/ab10_$nb.jsonl > $O/ab10_$nb.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for ln in open("$O/ab10_$nb.jsonl"):
    j = json.loads(ln)
    if "error" in j: print("%-18s ERROR %s" % (j["name"], j["error"][:160])); continue
    print("%-18s %8.1f ms %7.1f MB/s ok=%s same=%s %s" % (j["name"], j["code_ms"], j["MBps"], j["status_ok"], j["same_bytes_as_first"], j["origin"][:22]))
PY
done
stamp
line() {
  python - <<PY | tee -a $O/summary.txt
import json
try:
    j = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    a = j.get("api") or {}
    print("%-10s value=%7.1f MB/s code_ms=%8.1f ok=%s verified=%s api=%s %s" % ("$1", j["value"], j["kernel_ms"]["code"], j["all_status_ok"], j["roundtrip_verified_blocks"], a.get("value"), a.get("ms")))
except Exception as e:
    print("$1 FAILED", e, open("$O/bench_$1.err").read()[-400:])
PY
}
timeout 300 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --cpu-seconds 0 --steps 3 --warmup 1 > $O/bench_configs1.json 2> $O/bench_configs1.err; line configs1; stamp
timeout 400 python bench.py --kind mixed --cpu-seconds 0 --api-blocks 0 > $O/bench_mixed.json 2> $O/bench_mixed.err; line mixed; stamp
