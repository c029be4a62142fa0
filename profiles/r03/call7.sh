#!/bin/bash
# Round 3, GPU call 7: the round's numbers -- new tests, headline / decode / configs[1] / configs[3] bench lines, HBM counters
# on a sample of the headline, the north-star sweep with the CPU beside every line, the kernel trace of the headline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c7
mkdir -p $O
cd $R
T0=$(date +%s)
stamp() { echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt; }
echo "== new tests" | tee $O/summary.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "state_budget or two_engines" > $O/tests.txt 2>&1
tail -3 $O/tests.txt | tee -a $O/summary.txt; stamp
line() {  # name
  python - <<PY | tee -a $O/summary.txt
import json
try:
    j = json.loads(open("$O/bench_$1.json").read().strip().splitlines()[-1])
    a = j.get("api") or {}; c = j.get("cpu_baseline") or {}
    print("%-10s value=%7.1f MB/s code_ms=%8.1f frac=%.4f ok=%s verified=%s api=%s (first call %s ms) cpu=%s x%s" % ("$1", j["value"], j["kernel_ms"]["code"], j["roofline"]["frac"],
          j["all_status_ok"], j["roundtrip_verified_blocks"], a.get("value"), (a.get("ms") or {}).get("library_total_first_call"), c.get("value"), c.get("cores")))
except Exception as e:
    print("$1 FAILED", e, open("$O/bench_$1.err").read()[-400:])
PY
}
timeout 500 python bench.py --cpu-seconds 10 > $O/bench_headline.json 2> $O/bench_headline.err; line headline; stamp
timeout 500 python bench.py --mode decode --cpu-seconds 10 --api-blocks 0 > $O/bench_decode.json 2> $O/bench_decode.err; line decode; stamp
timeout 300 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --cpu-seconds 10 --steps 3 --warmup 1 > $O/bench_configs1.json 2> $O/bench_configs1.err; line configs1; stamp
timeout 400 python bench.py --kind mixed --cpu-seconds 0 --api-blocks 0 > $O/bench_mixed.json 2> $O/bench_mixed.err; line mixed; stamp
echo "== north-star sweep" | tee -a $O/summary.txt
timeout 600 python profiles/sweep_north.py $O/sweep_north.jsonl > $O/sweep.log 2>&1
python - <<PY | tee -a $O/summary.txt
import json
for ln in open("$O/sweep_north.jsonl"):
    j = json.loads(ln)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.items() if k in ("block_bytes", "blocks", "kind", "MBps", "code_ms", "roofline_frac", "cpu_MBps", "vs_cpu", "ok", "decoded_back", "block0_identical_to_reference", "error", "skipped", "state_GiB")})
PY
stamp
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace of the headline" | tee -a $O/summary.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1 > $O/prof.log 2>&1
ls $O/prof/*/ 2>/dev/null | head; find $O/prof -name "*kernel_stats.csv" -exec head -12 {} \; | tee -a $O/summary.txt
find $O/prof -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
stamp
echo "== HBM counters on a sample of the headline (first 48 KiB of every 1 MiB block: same chain, tables, code object)" | tee -a $O/summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/profiles/pmc_driver.py 1024 1048576 49152 > $O/pmc_$c.log 2>&1
  echo "$c rc=$?"; grep compressed $O/pmc_$c.log
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
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
stamp
