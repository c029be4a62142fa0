#!/bin/bash
# GPU call 17 of round 5: HBM counters of the persistent encoder launch (ONE dispatch: whole 1024 x 1 MiB passes finish), kernel
# trace of the default bench command, the 8-engine test again, the archiver with the fixed extract patch, the north-star sweep
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_engines or persistent_launch_gives_up or input_tail" > $O/c17_tests.txt 2>&1
tail -5 $O/c17_tests.txt
timeout 600 python profiles/r05/cli_bench.py --files 256 --unpatched-threads "" --out $O/c17_cli.json > $O/c17_cli.log 2>&1
cut -c1-220 $O/c17_cli.log
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/profiles/pmc_driver.py 1024 1048576 > $O/c17_pmc_plain.log 2>&1; tail -1 $O/c17_pmc_plain.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/c17_pmc_$c -o p -- python $R/profiles/pmc_driver.py 1024 1048576 > $O/c17_pmc_$c.log 2>&1
  echo "$c rc=$?"; tail -1 $O/c17_pmc_$c.log
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/c17_trace -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 > $O/c17_trace_bench.json 2> $O/c17_trace_bench.err
echo "trace rc=$?"
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/c17_pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
for f in glob.glob("$O/c17_trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
find $O -name "*.db" -delete 2>/dev/null
timeout 900 python profiles/sweep_north.py $O/c17_sweep_north.jsonl > $O/c17_sweep.log 2>&1
python - <<PY
import json
for ln in open("$O/c17_sweep_north.jsonl"):
    j = json.loads(ln)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.items() if k in ("block_bytes", "blocks", "kind", "MBps", "code_ms", "roofline_frac", "cpu_MBps", "vs_cpu", "ok", "decoded_back", "blocks_identical_to_reference", "block0_identical_to_reference", "error", "skipped")})
PY
