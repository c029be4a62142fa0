#!/bin/bash
# GPU call 4 of round 6: counters on the final code.  (a) WRITE_SIZE over the persistent launch (whole 1024 x 1 MiB sequence);
# (b) read side ON THE PERSISTENT LAUNCH ITSELF: FETCH_SIZE / TCC_EA0_RDREQ_sum on a small grid (256 blocks x 32 KiB, throughput
# shape) and on the full grid -- the arrival handshake of this round says whether the counted dispatch gets its workgroups
# resident at all; (c) FETCH_SIZE over the step kernels of the same code object (round 5's proxy, for comparison);
# (d) the lockstep decoder on BOTH chains of the mixed corpus (text n = 23, records n = 29); (e) kernel trace of the default line
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
export ZPAQ_AMD_LOG=1
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_api.py -m gpu -x -q -k "prefix_on_the_device or shelved_forms or foreign_kernel or legacy_models_at_baseline" -s > $O/c4_tests.txt 2>&1
tail -6 $O/c4_tests.txt; grep -E "first 16 KiB|given up" $O/c4_tests.txt
env -u ZPAQ_AMD_LOG timeout 1500 python bench.py > $O/c4_bench_default.json 2> $O/c4_bench_default.err
env -u ZPAQ_AMD_LOG timeout 900 python bench.py --kind mixed --configs1 0 --legacy 0 > $O/c4_bench_mixed.json 2> $O/c4_bench_mixed.err
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/profiles/pmc_driver.py 1024 1048576 > $O/c4_plain.log 2>&1; tail -2 $O/c4_plain.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4_pmc_persist_WRITE_SIZE -o p -- python $R/profiles/pmc_driver.py 1024 1048576 > $O/c4_pmc_persist_WRITE_SIZE.log 2>&1
echo "write rc=$?"; tail -3 $O/c4_pmc_persist_WRITE_SIZE.log
for c in FETCH_SIZE TCC_EA0_RDREQ_sum; do
  ZPAQ_AMD_PIPE_MODE=throughput ZPAQ_AMD_PERSIST_TIMEOUT_MS=15000 timeout 150 rocprofv3 --pmc $c --output-format csv -d $O/c4_pmc_small_$c -o p -- python $R/profiles/pmc_driver.py 256 1048576 32768 > $O/c4_pmc_small_$c.log 2>&1
  echo "small $c rc=$?"; grep -E "compressed|zpaq_amd" $O/c4_pmc_small_$c.log | tail -4
done
ZPAQ_AMD_PERSIST_TIMEOUT_MS=20000 timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_pmc_full_FETCH_SIZE -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c4_pmc_full_FETCH_SIZE.log 2>&1
echo "full fetch rc=$?"; grep -E "compressed|zpaq_amd" $O/c4_pmc_full_FETCH_SIZE.log | tail -4
ZPAQ_AMD_PIPE_PERSIST=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_pmc_steps_FETCH_SIZE -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c4_pmc_steps_FETCH_SIZE.log 2>&1
echo "steps fetch rc=$?"; grep compressed $O/c4_pmc_steps_FETCH_SIZE.log
for kind in text records; do
  timeout 300 python $R/profiles/pmc_decode_driver.py make 2048 1048576 131072 /tmp/zpq_dec_$kind.npz $kind > $O/c4_dec_make_$kind.log 2>&1; tail -2 $O/c4_dec_make_$kind.log
  timeout 200 python $R/profiles/pmc_decode_driver.py run /tmp/zpq_dec_$kind.npz > $O/c4_dec_plain_$kind.log 2>&1; tail -1 $O/c4_dec_plain_$kind.log
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/c4_pmc_dec_${kind}_$c -o p -- python $R/profiles/pmc_decode_driver.py run /tmp/zpq_dec_$kind.npz > $O/c4_pmc_dec_${kind}_$c.log 2>&1
    echo "dec $kind $c rc=$?"; grep decoded $O/c4_pmc_dec_${kind}_$c.log
  done
done
unset ZPAQ_AMD_LOG
timeout 900 rocprofv3 --kernel-trace --stats -d $O/c4_trace -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 > $O/c4_trace_bench.json 2> $O/c4_trace_bench.err
echo "trace rc=$?"
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/c4_pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
for f in glob.glob("$O/c4_trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:2500])
PY
find $O -name "*.db" -delete 2>/dev/null
tail -c 600 $O/c4_trace_bench.json
python - <<PY
import json
for f in ("c4_bench_default", "c4_bench_mixed"):
    try:
        j = json.loads([l for l in open("$O/" + f + ".json") if l.startswith("{")][-1])
        print(f, round(j["value"], 1), "ok", j["all_status_ok"], "frac", round(j["roofline"]["frac"], 4), "api", (j.get("api") or {}).get("value"), "ident", (j.get("reference_identity") or {}).get("identical"),
              "decode", (j.get("decode") or {}).get("value"), (j.get("decode") or {}).get("every_byte_verified"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
              "configs1", (j.get("configs1") or {}).get("value"), (j.get("configs1") or {}).get("error"),
              "legacy2", (j.get("legacy2") or {}).get("value"), ((j.get("legacy2") or {}).get("reference_identity") or {}).get("identical"), (j.get("legacy2") or {}).get("error"),
              "legacy3", (j.get("legacy3") or {}).get("value"), ((j.get("legacy3") or {}).get("reference_identity") or {}).get("identical"), (j.get("legacy3") or {}).get("error"))
    except Exception as e:
        print(f, "unreadable", e)
PY
