#!/bin/bash
# GPU call 1 of round 5: HBM counters of the lockstep decoder (one dispatch: a PMC pass finishes), separate passes for
# FETCH_SIZE and WRITE_SIZE, on the bench's decode chain: 2048 blocks, the first 128 KiB of each.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
timeout 100 python $R/profiles/pmc_decode_driver.py make 2048 1048576 131072 /tmp/zpq_dec.npz > $O/pmc_dec_make.log 2>&1
tail -2 $O/pmc_dec_make.log
timeout 60 python $R/profiles/pmc_decode_driver.py run /tmp/zpq_dec.npz > $O/pmc_dec_plain.log 2>&1
tail -1 $O/pmc_dec_plain.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 90 rocprofv3 --pmc $c --output-format csv -d $O/pmc_dec_$c -o p -- python $R/profiles/pmc_decode_driver.py run /tmp/zpq_dec.npz > $O/pmc_dec_$c.log 2>&1
  echo "$c rc=$?"; grep decoded $O/pmc_dec_$c.log
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/pmc_dec_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
find $O -name "*.db" -delete 2>/dev/null
# the decode leg on configs[4]'s own corpus (mixed text / LCG / records), round 4's code: the baseline of this round
cd $R
timeout 400 python bench.py --mode decode --kind mixed --cpu-seconds 0 > $O/decode_mixed_base.json 2> $O/decode_mixed_base.err
tail -c 1500 $O/decode_mixed_base.json; tail -5 $O/decode_mixed_base.err
