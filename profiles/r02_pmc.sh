#!/bin/bash
# HBM traffic of the pipelined encoder from PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE, as the MI355X guide
# prescribes, through a torch-free driver (rocprofv3 --pmc segfaults in processes that import torch on this image).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "small 64 65536" "headline 1024 1048576"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$1_$c
    timeout 1200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$1_$c -o p -- python $R/profiles/pmc_driver.py $2 $3 > $O/pmc_$1_$c.log 2>&1
    echo "$1 $c rc=$?"; grep compressed $O/pmc_$1_$c.log
  done
done
cd $R
python - <<PY > $O/pmc_summary.txt 2>&1
import csv, glob, collections
for cfg in ("small", "headline"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = collections.defaultdict(float); cnt = collections.Counter()
        for f in glob.glob("$O/pmc_%s_%s/**/*counter_collection.csv" % (cfg, c), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"][:48]
                tot[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k, v in sorted(tot.items()):
            print(cfg, c, k, "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
find $O -name "*counter_collection.csv" -size +1M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
cat $O/pmc_summary.txt
tail -4 $O/pmc_small_FETCH_SIZE.log
