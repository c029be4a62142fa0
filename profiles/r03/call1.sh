#!/bin/bash
# Round 3, GPU call 1: the units built at the end of round 2 run on the MI355X for the first time.
#   1. parity of the bit-lane / nibble-lane units (the tests VERDICT r02 names) with every knob on
#   2. in-process A/B of every knob on the headline (profiles/ab_inproc.py: one corpus, a fresh plan per variant)
#   3. where the wavefronts ran (-DZPQ_TRACE) and every unit alone (ZPAQ_AMD_PIPE_PROFILE), default vs all, 1024 x 64 KiB
#   4. cycle breakdown of the decoder (-DZPQ_PROF)
#   5. configs[1] with the ROW variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c1
mkdir -p $O
cd $R
export ZPAQ_AMD_MAX_JIT=256
T0=$(date +%s)
echo "== parity, all bit-lane units on" | tee $O/summary.txt
ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 timeout 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -x \
  -k "test_encode_matches_oracle_and_golden[4] or test_all_nine_component_types[4] or large_batch or mixed_corpus or records_block or ragged or legacy_min_mid_max_models[0-4] or legacy_min_mid_max_models[1-4] or legacy_min_mid_max_models[2-4] or random_hcomp" > $O/parity_all.txt 2>&1
tail -4 $O/parity_all.txt | tee -a $O/summary.txt
echo "== parity, map ILP 2 + full squash + row flat" | tee -a $O/summary.txt
ZPAQ_AMD_PIPE_MAP_ILP=2 ZPAQ_AMD_PIPE_FULL_SQUASH=1 ZPAQ_AMD_PIPE_ROW_FLAT=1 timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x \
  -k "test_encode_matches_oracle_and_golden[4] or test_all_nine_component_types[4] or legacy_min_mid_max_models[2-4]" > $O/parity_ilp.txt 2>&1
tail -4 $O/parity_ilp.txt | tee -a $O/summary.txt
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
echo "== A/B headline" | tee -a $O/summary.txt
timeout 900 python profiles/ab_inproc.py profiles/r03/ab1_headline.json --out $O/ab1.jsonl > $O/ab1.log 2>&1
python - <<PY | tee -a $O/summary.txt
import json
for ln in open("$O/ab1.jsonl"):
    j = json.loads(ln)
    if "error" in j: print("%-18s ERROR %s" % (j["name"], j["error"][:160])); continue
    print("%-18s %8.1f ms %7.1f MB/s ok=%s same=%s kind=%d" % (j["name"], j["code_ms"], j["MBps"], j["status_ok"], j["same_bytes_as_first"], j["kernel_kind"]))
PY
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
B64="python bench.py --blocks 1024 --block-bytes 65536 --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1"
ALL="ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1"
for v in default all; do
  E=""; [ $v = all ] && E="$ALL"
  env $E ZPAQ_AMD_SPEC_DEFS=-DZPQ_TRACE ZPAQ_AMD_PIPE_TRACE=$O/trace_$v.bin timeout 300 $B64 > $O/trace_$v.json 2> $O/trace_$v.err
  echo "== placement $v"; python profiles/pipe_trace.py $O/trace_$v.bin 2>&1 | head -24
  python - <<PY
import numpy as np
a = np.fromfile("$O/trace_$v.bin", dtype=np.uint64).reshape(-1, 4)
np.savez_compressed("$O/trace_$v.npz", a=a[:min(len(a), 400000)])
PY
  rm -f $O/trace_$v.bin
done 2>&1 | tee -a $O/summary.txt
for v in default all; do
  E=""; [ $v = all ] && E="$ALL"
  echo "== every unit alone, $v"
  env $E ZPAQ_AMD_PIPE_PROFILE=1 timeout 300 $B64 2>&1 >/dev/null | grep "pipe profile" | awk '{print $4, $5, $6, $7, $8}' | sort | uniq -c | sort -k2,2 -k3,3n | head -70
done 2>&1 | tee -a $O/summary.txt
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
echo "== decoder cycle breakdown (-DZPQ_PROF), 1024 x 64 KiB" | tee -a $O/summary.txt
ZPAQ_AMD_SPEC_DEFS=-DZPQ_PROF timeout 300 python bench.py --mode decode --blocks 1024 --block-bytes 65536 --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1 --kernel 3 > $O/dec_prof.json 2> $O/dec_prof.err
grep -h "spec prof" $O/dec_prof.json $O/dec_prof.err | tee -a $O/summary.txt
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
echo "== configs[1]" | tee -a $O/summary.txt
BENCH="python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --cpu-seconds 0 --steps 3 --warmup 1"
run() {
  local name=$1; shift
  env "$@" timeout 300 $BENCH > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("%-22s value=%7.1f MB/s code_ms=%8.1f ok=%s api=%s" % ("$name", j["value"], j["kernel_ms"]["code"], j["all_status_ok"], (j.get("api") or {}).get("value")))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-300:])
PY
}
run c1_default
run c1_rows_d3 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=3
run c1_rows_sq_ilp2 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=3 ZPAQ_AMD_PIPE_FULL_SQUASH=1 ZPAQ_AMD_PIPE_MAP_ILP=2
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
