#!/bin/bash
# Round 3, GPU call 6: the cleaned-up encoder (two shapes chosen by batch size, padded MIX rows, 8 hardware queues):
# parity of both shapes, the crossover, the headline bench, and HBM transactions of the headline from PMC counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c6
mkdir -p $O
cd $R
T0=$(date +%s)
echo "== parity of both shapes" | tee $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "both_shapes or zeros_known or method_3_known" > $O/parity.txt 2>&1
tail -3 $O/parity.txt | tee -a $O/summary.txt
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
for nb in 64 256 384 512 768 1024; do
  echo "== $nb blocks" | tee -a $O/summary.txt
  timeout 400 python profiles/ab_inproc.py profiles/r03/ab6_$nb.json --out $O/ab6_$nb.jsonl > $O/ab6_$nb.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for ln in open("$O/ab6_$nb.jsonl"):
    j = json.loads(ln)
    if "error" in j: print("%-18s ERROR %s" % (j["name"], j["error"][:160])); continue
    print("%-18s %8.1f ms %7.1f MB/s ok=%s same=%s" % (j["name"], j["code_ms"], j["MBps"], j["status_ok"], j["same_bytes_as_first"]))
PY
done
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
echo "== headline bench" | tee -a $O/summary.txt
timeout 600 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
python - <<PY | tee -a $O/summary.txt
import json
j = json.loads(open("$O/bench_headline.json").read().strip().splitlines()[-1])
print("value %.1f MB/s code_ms %.1f frac %.4f api %s cpu %s" % (j["value"], j["kernel_ms"]["code"], j["roofline"]["frac"], (j.get("api") or {}).get("value"), (j.get("cpu_baseline") or {}).get("value")))
print(j["roofline"]["kernel_origin"])
PY
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\(sum\)\?" | sort -u | head -80 > $O/tcc_counters.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_calib_$c -o p -- $R/profiles/r03/gups --calib > $O/pmc_calib_$c.log 2>&1
  echo "calib $c rc=$?"
done 2>&1 | tee -a $O/summary.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 400 rocprofv3 --pmc $c --kernel-include-regex "zpq_pipe|init_arena" --kernel-iteration-range "[1000-1031]" --output-format csv -d $O/pmc_head_$n -o p -- python $R/profiles/pmc_driver.py 1024 1048576 > $O/pmc_head_$n.log 2>&1
  echo "headline $c rc=$?"; grep compressed $O/pmc_head_$n.log
done 2>&1 | tee -a $O/summary.txt
cd $R
python - <<PY | tee -a $O/summary.txt
import csv, glob, collections
for d in sorted(glob.glob("$O/pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v, "per_dispatch=%.1f" % (v / cnt[k]))
PY
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
