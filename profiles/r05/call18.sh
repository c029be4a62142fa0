#!/bin/bash
# GPU call 18 of round 5: where configs[1]'s unit wavefronts spend their time; FETCH_SIZE of the persistent encoder again (sampled: the first 96 KiB of every block)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
ZPAQ_AMD_PERSIST_PROF=$O/c18_prof_configs1.bin timeout 300 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --cpu-seconds 0 --api-blocks 0 > $O/c18_configs1.json 2> $O/c18_configs1.err
python -c "import json; d=json.load(open('$O/c18_configs1.json')); print('configs1', round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
python profiles/persist_prof.py $O/c18_prof_configs1.bin | head -14
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/c18_pmc_$c -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c18_pmc_$c.log 2>&1
  echo "$c rc=$?"; grep compressed $O/c18_pmc_$c.log
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/c18_pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
find $O -name "*.db" -delete 2>/dev/null
