#!/bin/bash
# GPU call 9 of round 4: (1) HBM counters of the headline in STEADY STATE: FETCH_SIZE / WRITE_SIZE (separate passes) of steps
# 1000-1099 of each encoder kernel -- the middle of every block, tables warm -- with the rest of the sequence running unprofiled
# (rocprofv3 --kernel-iteration-range); (2) what the archiver's 256 threads hand the library (batch log).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  (time timeout 260 rocprofv3 --pmc $c --kernel-include-regex "zpq_pipe_.*" --kernel-iteration-range "[1000-1099]" --output-format csv -d $O/pmc_$c -o p -- python $R/profiles/pmc_driver.py 1024 1048576) > $O/pmc_$c.log 2>&1
  echo "$c rc=$?"; grep -a compressed $O/pmc_$c.log; tail -3 $O/pmc_$c.log
done
cd $R
python - <<PY > $O/pmc_steady_summary.txt 2>&1
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(c, k, "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
cat $O/pmc_steady_summary.txt
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +4M -delete 2>/dev/null
# (2)
W=/tmp/zpq_probe; rm -rf $W; mkdir -p $W/tree
python - <<PY
import sys; sys.path.insert(0, '.')
from zpaq_amd import corpus
for i in range(160):
    corpus.block("text", 1 << 20, 777 + i).tofile(f"$W/tree/f{i:03d}.txt")
PY
cd $W
( time ZPAQ_AMD_LOG=1 timeout 120 $R/oracle/_ref/zpaq_amd_cli add ours.zpaq tree -method 50 -threads 256 ) > add.log 2>&1
grep -aF "[zpaq_amd]" add.log | cut -c1-230 | head -40; grep -aE "seconds|real" add.log
cp add.log $O/cli_probe_256.log
