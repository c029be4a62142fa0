#!/bin/bash
# First GPU call of the next round: A/B of the bit-lane units built at the end of round 2 (emulator-exact, never run on
# the MI355X; off by default).  Parity first (a wrong kernel is not worth timing), then the headline with each knob.
#   bash profiles/r03_prebuild_variants.sh          (here, CPU: the variants' code objects, so the GPU call never waits for hipRTC)
#   gpurun --timeout 2700 -- 'bash profiles/r03_ab_bits.sh'          (about 45 GPU-minutes: ~40 bench runs of ~45 s, parity, two profiles,
#                                                                    two traces; every step has its own timeout; summary in gpurun_out/r03ab/summary.txt)
# Knobs (host/codegen.cpp, part of the generated source and therefore of the cache key; unseen variants go through hipRTC):
#   ZPAQ_AMD_PIPE_MIX_BITS=1     MIX with a lane per (block, bit position, weight quad)   ZPAQ_AMD_PIPE_MIX_DEPTH=1..4 (3)
#   ZPAQ_AMD_PIPE_LIGHT_BITS=m   1 CM | 2 MIX2 | 4 SSE with a lane per (block, bit position)   ZPAQ_AMD_PIPE_LIGHT_DEPTH=1..4 (3)
#   ZPAQ_AMD_PIPE_ROW_NIBBLES=1  ROW units with a lane per (block, nibble)                  ZPAQ_AMD_PIPE_ROW_DEPTH=1..4 (2)
#   ZPAQ_AMD_PIPE_ROW_FLAT=1     the one-lane ROW unit with the candidate row picked by masks, not branches (-17 % instructions)
#   ZPAQ_AMD_PIPE_MAP_ILP=2|4    ICM / ISSE maps with 2 / 4 blocks per lane: independent chains interleaved in one wavefront
#   ZPAQ_AMD_PIPE_FULL_SQUASH=1  squash from the whole table in LDS (5 instructions fewer per bit in ISSE / MIX / MIX2 / coder)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03ab
mkdir -p $O
cd $R
export ZPAQ_AMD_MAX_JIT=256
PAR="tests/test_gpu_parity.py -m gpu -q -x -k 'nine or legacy or large or records or mixed or ragged or zeros or compress_blocks'"
echo "== parity with every bit-lane unit on" | tee $O/summary.txt
ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 timeout 1200 bash -c "python -m pytest $PAR" > $O/parity_bits.txt 2>&1
tail -3 $O/parity_bits.txt | tee -a $O/summary.txt
BENCH="python bench.py --cpu-seconds 0 --api-blocks 0 --steps 1 --warmup 1"
run() {   # name, env...
  local name=$1; shift
  env "$@" timeout 600 $BENCH > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("%-28s value=%7.1f MB/s code_ms=%8.1f frac=%.4f ok=%s verified=%s origin=%s" % ("$name", j["value"], j["kernel_ms"]["code"],
          j["roofline"]["frac"], j["all_status_ok"], j["roundtrip_verified_blocks"], j["roofline"]["kernel_origin"][:14]))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-400:])
PY
}
run default
for d in 1 2 3 4; do run mix_d$d ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_MIX_DEPTH=$d; done
for d in 1 2 3; do run rows_d$d ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=$d; done
run full_squash ZPAQ_AMD_PIPE_FULL_SQUASH=1
run row_flat ZPAQ_AMD_PIPE_ROW_FLAT=1
run map_ilp2 ZPAQ_AMD_PIPE_MAP_ILP=2
run map_ilp4 ZPAQ_AMD_PIPE_MAP_ILP=4
run map_ilp2_squash ZPAQ_AMD_PIPE_MAP_ILP=2 ZPAQ_AMD_PIPE_FULL_SQUASH=1
run light_cm ZPAQ_AMD_PIPE_LIGHT_BITS=1
run light_mix2 ZPAQ_AMD_PIPE_LIGHT_BITS=2
run light_sse ZPAQ_AMD_PIPE_LIGHT_BITS=4
run light_all ZPAQ_AMD_PIPE_LIGHT_BITS=7
run mix_rows ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_ROW_NIBBLES=1
run all_ilp2 ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_MAP_ILP=2 ZPAQ_AMD_PIPE_FULL_SQUASH=1
for d in 2 3 4; do run all_d$d ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_MIX_DEPTH=$d ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_LIGHT_DEPTH=$d ZPAQ_AMD_PIPE_ROW_NIBBLES=1; done
# fewer, longer steps once the kernels are short (launch / event overhead per step is fixed)
run all_d3_c1024 ZPAQ_AMD_PIPE_CHUNK=1024 ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1
run default_c1024 ZPAQ_AMD_PIPE_CHUNK=1024
# BASELINE configs[1] (-m3, 256 x 256 KiB LCG: ICM + ISSE, 256 lanes per unit -- the latency-bound small-batch regime, where
# the fetch depth of the ROW units should matter most)
BENCH="python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --cpu-seconds 0 --steps 3 --warmup 1"
run c1_default
for d in 2 3 4; do run c1_rows_d$d ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=$d; done
run c1_rows_squash ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=3 ZPAQ_AMD_PIPE_FULL_SQUASH=1
run c1_rows_ilp2 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=3 ZPAQ_AMD_PIPE_FULL_SQUASH=1 ZPAQ_AMD_PIPE_MAP_ILP=2
BENCH="python bench.py --cpu-seconds 0 --api-blocks 0 --steps 1 --warmup 1"
# per-kernel durations of the best candidate and of the default, for the timeline
cd /tmp && export TMPDIR=/tmp
for v in default all; do
  E=""; [ $v = all ] && E="ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1"
  env $E timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o p -- python $R/bench.py --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1 > $O/prof_$v.log 2>&1
  python $R/profiles/pipe_timeline.py $O/prof_$v/p_results.db > $O/timeline_$v.txt 2>&1
  rm -f $O/prof_$v/*.db
  echo "== timeline $v"; head -12 $O/timeline_$v.txt
done | tee -a $O/summary.txt
# where every wavefront of a step ran and for how long (kernels built -DZPQ_TRACE; records -> profiles/pipe_trace.py)
cd $R
for v in default all; do
  E=""; [ $v = all ] && E="ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1"
  env $E ZPAQ_AMD_SPEC_DEFS=-DZPQ_TRACE ZPAQ_AMD_PIPE_TRACE=$O/trace_$v.bin timeout 600 python bench.py --blocks 1024 --block-bytes 65536 \
      --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1 > $O/trace_$v.json 2> $O/trace_$v.err
  echo "== placement $v"; python profiles/pipe_trace.py $O/trace_$v.bin 2>&1 | head -24
  rm -f $O/trace_$v.bin
done | tee -a $O/summary.txt
# host-buffer API with the page-locked staging buffer (ZPAQ_AMD_PINNED_STAGE=1, experimental): the `api` object of bench.py
cd $R
for v in pageable pinned; do
  E=""; [ $v = pinned ] && E="ZPAQ_AMD_PINNED_STAGE=1"
  env $E timeout 600 python bench.py --cpu-seconds 0 --steps 1 --warmup 1 > $O/api_$v.json 2> $O/api_$v.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    j = json.loads(open("$O/api_$v.json").read().strip().splitlines()[-1])
    print("api %-9s %.1f MB/s  ms=%s" % ("$v", j["api"]["value"], {k: round(x, 1) for k, x in j["api"]["ms"].items()}))
except Exception as e:
    print("api $v FAILED", e, open("$O/api_$v.err").read()[-400:])
PY
done
