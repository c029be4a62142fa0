#!/bin/bash
# Round 3, GPU call 14: ROW units that do not write back an unchanged row -- parity, text / zeros timing, PMC of the new code object
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c14
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt; }
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_encode_matches_oracle_and_golden[4] or test_all_nine_component_types[4] or large_batch or zeros_known or mixed_corpus" > $O/tests.txt 2>&1
tail -3 $O/tests.txt | tee $O/summary.txt; stamp
for k in text zeros; do
  timeout 200 python profiles/ab_inproc.py profiles/r03/ab14_$k.json --out $O/ab14_$k.jsonl > $O/ab14_$k.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for ln in open("$O/ab14_$k.jsonl"):
    j = json.loads(ln)
    print("$k", j.get("code_ms"), j.get("MBps"), j.get("status_ok"), j.get("origin"), j.get("error"))
PY
done
stamp
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2 3; do
    rm -rf $O/pmc_$c
    timeout 60 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/profiles/pmc_driver.py 1024 1048576 49152 > $O/pmc_$c.log 2>&1
    rc=$?; echo "$c attempt $attempt rc=$rc"; [ $rc = 0 ] && break
  done
done 2>&1 | tee -a $O/summary.txt
cd $R
python - <<PY | tee -a $O/summary.txt
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float)
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Kernel_Name"][:24]] += float(r["Counter_Value"])
    for k, v in sorted(tot.items()):
        print(c, k, "sum=%.1f" % v)
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
stamp
