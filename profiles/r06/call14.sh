#!/bin/bash
# GPU call 14 of round 6: the final code (MATCH at three lines in the packing): GPU suite, default line, mixed, dense, sweep,
# counters on the new code object (FETCH_SIZE up to three attempts, WRITE_SIZE), kernel trace
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/c14_gputest.txt 2>&1
tail -14 $O/c14_gputest.txt
timeout 1500 python bench.py > $O/c14_bench_default.json 2> $O/c14_bench_default.err
timeout 900 python bench.py --kind mixed --configs1 0 --legacy 0 > $O/c14_bench_mixed.json 2> $O/c14_bench_mixed.err
timeout 900 python bench.py --blocks 2048 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0 > $O/c14_bench_dense.json 2> $O/c14_bench_dense.err
cd /tmp && export TMPDIR=/tmp
ZPAQ_AMD_LOG=1 ZPAQ_AMD_PERSIST_TIMEOUT_MS=20000 timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c14_pmc_persist_WRITE_SIZE -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c14_pmc_persist_WRITE_SIZE.log 2>&1
echo "persist WRITE_SIZE rc=$?"
for k in 1 2 3; do
  ZPAQ_AMD_LOG=1 ZPAQ_AMD_PERSIST_TIMEOUT_MS=20000 timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c14_pmc_persist_FETCH_SIZE_$k -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c14_pmc_persist_FETCH_SIZE_$k.log 2>&1
  rc=$?; echo "persist FETCH_SIZE attempt $k rc=$rc"; [ $rc = 0 ] && break
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/c14_trace -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 --configs1 0 --legacy 0 > $O/c14_trace_bench.json 2> $O/c14_trace_bench.err
echo "trace rc=$?"
cd $R
python - <<PY
import csv, glob, collections, json
for d in sorted(glob.glob("$O/c14_pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
for f in glob.glob("$O/c14_trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:700])
for f in ("c14_bench_default", "c14_bench_mixed", "c14_bench_dense"):
    try:
        j = json.loads([l for l in open("$O/" + f + ".json") if l.startswith("{")][-1])
        print(f, round(j["value"], 1), "ok", j["all_status_ok"], "frac", round(j["roofline"]["frac"], 4), "origin", j["roofline"]["kernel_origin"][:26], "traffic", j["roofline"]["traffic"], "api", (j.get("api") or {}).get("value"), (j.get("api") or {}).get("persistent_launch"), "ident", (j.get("reference_identity") or {}).get("identical"),
              "decode", (j.get("decode") or {}).get("value"), (j.get("decode") or {}).get("every_byte_verified"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
              "configs1", (j.get("configs1") or {}).get("value"), "legacy2", (j.get("legacy2") or {}).get("value"), ((j.get("legacy2") or {}).get("reference_identity") or {}).get("identical"),
              "legacy3", (j.get("legacy3") or {}).get("value"), ((j.get("legacy3") or {}).get("reference_identity") or {}).get("identical"), (j.get("legacy3") or {}).get("error"))
    except Exception as e:
        print(f, "unreadable", e)
PY
find $O -name "*.db" -delete 2>/dev/null
timeout 1200 python profiles/sweep_north.py $O/c14_sweep_north.jsonl > $O/c14_sweep.log 2>&1
python - <<PY
import json
for ln in open("$O/c14_sweep_north.jsonl"):
    j = json.loads(ln)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.items() if k in ("block_bytes", "blocks", "kind", "MBps", "roofline_frac", "cpu_MBps", "ok", "decoded_back", "blocks_identical_to_reference", "error", "skipped")})
PY
