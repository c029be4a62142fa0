#!/bin/bash
# Round 3, GPU call 8: suffix arrays on the device, the 16 MiB known answer, WRITE_SIZE on the headline sample
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c8
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt; }
echo "== tests" | tee $O/summary.txt
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "suffix_arrays or 16_mib or method_3_known or pcomp_post" > $O/tests.txt 2>&1
tail -5 $O/tests.txt | tee -a $O/summary.txt; stamp
timeout 300 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --cpu-seconds 0 --steps 3 --warmup 1 > $O/bench_configs1.json 2> $O/bench_configs1.err
python - <<PY | tee -a $O/summary.txt
import json
j = json.loads(open("$O/bench_configs1.json").read().strip().splitlines()[-1])
print("configs1 value=%.1f api=%s ms=%s" % (j["value"], j["api"]["value"], j["api"]["ms"]))
PY
stamp
cd /tmp && export TMPDIR=/tmp
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE_SIZE -o p -- python $R/profiles/pmc_driver.py 1024 1048576 49152 > $O/pmc_WRITE_SIZE.log 2>&1
echo "WRITE_SIZE rc=$?" | tee -a $O/summary.txt; grep compressed $O/pmc_WRITE_SIZE.log | tee -a $O/summary.txt
cd $R
python - <<PY | tee -a $O/summary.txt
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("$O/pmc_WRITE_SIZE/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:40], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k, v in sorted(tot.items()):
    print(k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
find $O -name "*.db" -delete 2>/dev/null
stamp
