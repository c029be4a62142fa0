#!/bin/bash
# GPU call 24 of round 6: the final code -- ROW halves' re-fetch without waiting for store acknowledgements; -m4 / mid.cfg /
# configs[1] again, the whole GPU suite, the default line, kernel trace, mixed, dense, sweep
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "small_chains or method_3 or legacy_models_at" > $O/c24_tests_small.txt 2>&1
tail -3 $O/c24_tests_small.txt
run() { name=$1; shift; timeout 400 python bench.py "$@" > $O/c24_$name.json 2> $O/c24_$name.err; }
Q="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16"
run m4_256 $Q --method 4 --blocks 256
run m4_1024 $Q --method 4 --blocks 1024
run legacy2 --legacy-level 2 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0
run configs1_quick --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0
ZPAQ_AMD_PERSIST_PROF=$O/c24_prof_m4_256.bin timeout 300 python bench.py $Q --method 4 --blocks 256 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c24_prof_m4_256.bin > $O/c24_prof_m4_256.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/c24_gputest.txt 2>&1
tail -14 $O/c24_gputest.txt
timeout 1500 python bench.py > $O/c24_bench_default.json 2> $O/c24_bench_default.err
timeout 600 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 > $O/c24_bench_configs1.json 2> $O/c24_bench_configs1.err
timeout 900 python bench.py --kind mixed --configs1 0 --legacy 0 > $O/c24_bench_mixed.json 2> $O/c24_bench_mixed.err
timeout 900 python bench.py --blocks 2048 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0 > $O/c24_bench_dense.json 2> $O/c24_bench_dense.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/c24_trace -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 --legacy 0 > $O/c24_trace_bench.json 2> $O/c24_trace_bench.err
echo "trace rc=$?"
cd $R
python - <<PY
import json, glob
for f in glob.glob("$O/c24_trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:900])
for f in sorted(glob.glob("$O/c24_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "ok", j["all_status_ok"], "frac", round(j["roofline"]["frac"], 4), "origin", j["roofline"]["kernel_origin"][:20], "traffic", j["roofline"]["traffic"], "api", (j.get("api") or {}).get("value"), (j.get("api") or {}).get("persistent_launch"), "ident", (j.get("reference_identity") or {}).get("identical"),
              "decode", (j.get("decode") or {}).get("value"), (j.get("decode") or {}).get("every_byte_verified"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
              "configs1", (j.get("configs1") or {}).get("value"), ((j.get("configs1") or {}).get("api") or {}).get("value"), "legacy2", (j.get("legacy2") or {}).get("value"), ((j.get("legacy2") or {}).get("reference_identity") or {}).get("identical"),
              "legacy3", (j.get("legacy3") or {}).get("value"), ((j.get("legacy3") or {}).get("reference_identity") or {}).get("identical"), (j.get("legacy3") or {}).get("error"))
    except Exception as e:
        print(f, "unreadable", e)
PY
find $O -name "*.db" -delete 2>/dev/null
head -40 $O/c24_prof_m4_256.txt
