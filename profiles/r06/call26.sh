#!/bin/bash
# GPU call 26 of round 6: WRITE_SIZE over the final code object's persistent launch (call 25's pass did not return), up to four attempts
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 4; do
  ZPAQ_AMD_LOG=1 ZPAQ_AMD_PERSIST_TIMEOUT_MS=20000 timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c26_pmc_persist_WRITE_SIZE_$k -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c26_pmc_persist_WRITE_SIZE_$k.log 2>&1
  rc=$?; echo "persist WRITE_SIZE attempt $k rc=$rc"; [ $rc = 0 ] && break
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/c26_pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
find $O -name "*.db" -delete 2>/dev/null
