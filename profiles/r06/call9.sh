#!/bin/bash
# GPU call 9 of round 6: the final code.  The whole GPU suite, the default bench line as the driver runs it, the mixed corpus,
# configs[1], counters on the headline's code object (FETCH_SIZE and WRITE_SIZE over the persistent launch itself, full grid, the
# first 96 KiB of every block), the kernel trace of the default line's command, the north-star sweep
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/c9_gputest.txt 2>&1
tail -5 $O/c9_gputest.txt
timeout 1500 python bench.py > $O/c9_bench_default.json 2> $O/c9_bench_default.err
timeout 900 python bench.py --kind mixed --configs1 0 --legacy 0 > $O/c9_bench_mixed.json 2> $O/c9_bench_mixed.err
timeout 400 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 > $O/c9_bench_configs1.json 2> $O/c9_bench_configs1.err
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ZPAQ_AMD_LOG=1 ZPAQ_AMD_PERSIST_TIMEOUT_MS=20000 timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/c9_pmc_persist_$c -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c9_pmc_persist_$c.log 2>&1
  echo "persist $c rc=$?"; grep -E "compressed|zpaq_amd" $O/c9_pmc_persist_$c.log | tail -3
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/c9_trace -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 --configs1 0 --legacy 0 > $O/c9_trace_bench.json 2> $O/c9_trace_bench.err
echo "trace rc=$?"
cd $R
python - <<PY
import csv, glob, collections, json
for d in sorted(glob.glob("$O/c9_pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
for f in glob.glob("$O/c9_trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
for f in ("c9_bench_default", "c9_bench_mixed", "c9_bench_configs1"):
    try:
        j = json.loads([l for l in open("$O/" + f + ".json") if l.startswith("{")][-1])
        print(f, round(j["value"], 1), "ok", j["all_status_ok"], "frac", round(j["roofline"]["frac"], 4), "origin", j["roofline"]["kernel_origin"][:26], "api", (j.get("api") or {}).get("value"), "ident", (j.get("reference_identity") or {}).get("identical"),
              "decode", (j.get("decode") or {}).get("value"), (j.get("decode") or {}).get("every_byte_verified"), ((j.get("decode") or {}).get("roofline") or {}).get("traffic"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
              "configs1", (j.get("configs1") or {}).get("value"), (j.get("configs1") or {}).get("error"),
              "legacy2", (j.get("legacy2") or {}).get("value"), ((j.get("legacy2") or {}).get("reference_identity") or {}).get("identical"), (j.get("legacy2") or {}).get("error"),
              "legacy3", (j.get("legacy3") or {}).get("value"), ((j.get("legacy3") or {}).get("reference_identity") or {}).get("identical"), (j.get("legacy3") or {}).get("error"))
    except Exception as e:
        print(f, "unreadable", e)
PY
find $O -name "*.db" -delete 2>/dev/null
timeout 1200 python profiles/sweep_north.py $O/c9_sweep_north.jsonl > $O/c9_sweep.log 2>&1
python - <<PY
import json
for ln in open("$O/c9_sweep_north.jsonl"):
    j = json.loads(ln)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.items() if k in ("block_bytes", "blocks", "kind", "MBps", "code_ms", "roofline_frac", "cpu_MBps", "vs_cpu", "ok", "decoded_back", "blocks_identical_to_reference", "error", "skipped")})
PY
