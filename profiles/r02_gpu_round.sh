#!/bin/bash
# One GPU call: the -m gpu suite, the bench lines of the round, a kernel trace and the PMC traffic passes of the headline.
#   gpurun --timeout 3000 -- 'bash profiles/r02_gpu_round.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gputest.txt
python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --steps 3 > $O/bench_configs1.json 2> $O/bench_configs1.err
python bench.py --blocks 2048 --cpu-seconds 0 --api-blocks 0 > $O/bench_dense.json 2> $O/bench_dense.err
python bench.py --mode decode --cpu-seconds 0 --api-blocks 0 > $O/bench_decode.json 2> $O/bench_decode.err
python bench.py --kind mixed --blocks 1024 --cpu-seconds 0 --api-blocks 0 > $O/bench_mixed.json 2> $O/bench_mixed.err
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_headline -o headline -- $BENCH > $O/prof_headline.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "zpq" --output-format csv -d $O/pmc_fetch -o fetch -- $BENCH > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "zpq" --output-format csv -d $O/pmc_write -o write -- $BENCH > $O/pmc_write.log 2>&1
cd $R
python profiles/pipe_timeline.py $O/prof_headline/headline_results.db > $O/timeline_headline.txt 2>&1
python profiles/rocpd_summary.py $O/prof_headline/headline_results.db $O/r02_headline > /dev/null 2>&1
python - <<PY > $O/pmc_summary.txt 2>&1
import csv, glob, collections
for name in ("fetch", "write"):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            tot[(r["Kernel_Name"][:40], r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(r["Kernel_Name"][:40], r["Counter_Name"])] += 1
    for k, v in sorted(tot.items()):
        print(name, k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
rm -rf $O/pmc_fetch/*/*.db $O/prof_headline/*.db 2>/dev/null
find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
tail -3 $O/gputest.txt
for f in headline configs1 dense decode mixed; do python - <<PY
import json
try:
    j = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", "value=%.1f MB/s" % j["value"], "code_ms=%.1f" % j["kernel_ms"]["code"], "frac=%.4f" % j["roofline"]["frac"], "ok=", j["all_status_ok"],
          "api=", (j.get("api") or {}).get("value"), "cpu=", (j.get("cpu_baseline") or {}).get("value"), "identical=", (j.get("cpu_baseline") or {}).get("bit_identical_vs_reference"),
          (j.get("cpu_baseline") or {}).get("compared_how"))
except Exception as e:
    print("$f FAILED", e, open("$O/bench_$f.err").read()[-600:])
PY
done
cat $O/timeline_headline.txt | head -12
cat $O/pmc_summary.txt
