#!/bin/bash
# GPU call 20 of round 5: (a) where the patched archiver's wall clock goes (the library's line per device batch beside it);
# (b) read requests of the encoder: TCC_EA0_RDREQ_sum on the persistent launch (FETCH_SIZE, three counters, hung twice), and
# FETCH_SIZE on the step kernels of the same units (ZPAQ_AMD_PIPE_PERSIST=0), both on the first 96 KiB of every block
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 200 python profiles/r05/cli_bench.py --quick --out $O/c20_cli_quick.json > $O/c20_cli.log 2>&1
python -c "
import json
for r in json.load(open('$O/c20_cli_quick.json'))['rows']: print(r['what'][:60], round(r.get('wall_s',0),2), r.get('library_log'), r.get('archiver_says'))"
cd /tmp && export TMPDIR=/tmp
ZPAQ_AMD_PERSIST_TIMEOUT_MS=60000 timeout 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum --output-format csv -d $O/c20_pmc_rdreq -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c20_pmc_rdreq.log 2>&1
echo "rdreq rc=$?"; grep compressed $O/c20_pmc_rdreq.log
ZPAQ_AMD_PERSIST_TIMEOUT_MS=60000 timeout 150 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/c20_pmc_rdreq32 -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c20_pmc_rdreq32.log 2>&1
echo "rdreq32 rc=$?"; grep compressed $O/c20_pmc_rdreq32.log
ZPAQ_AMD_PIPE_PERSIST=0 timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c20_pmc_fetch_steps -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c20_pmc_fetch_steps.log 2>&1
echo "fetch steps rc=$?"; grep compressed $O/c20_pmc_fetch_steps.log
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/c20_pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
find $O -name "*.db" -delete 2>/dev/null
