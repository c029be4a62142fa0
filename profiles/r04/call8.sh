#!/bin/bash
# GPU call 8 of round 4: (1) where the API call's 250 ms beside the kernels go (batch log) with outputs through a pinned
# buffer; (2) lockstep decoder A/B: 32 row lanes per block (8 wavefronts, two per SIMD), mixers at a higher priority.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
(ZPAQ_AMD_LOG=1 timeout 200 python bench.py --cpu-seconds 0 --decode-blocks 0 --steps 1 --warmup 1 --verify-blocks 0) > gpurun_out/r04/bench_api_log.json 2> gpurun_out/r04/bench_api_log.err
grep -a "zpaq_amd\]" gpurun_out/r04/bench_api_log.err | tail -3
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_api_log.json", errors="replace") if l.startswith("{")][-1])
print("headline", round(d["value"], 1), "api", round(d["api"]["value"], 1), d["api"]["ms"])
PY
run() {
  n=$1; shift
  (env "$@" timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 262144 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0 --decode-blocks 0) > gpurun_out/r04/teamab_$n.json 2> gpurun_out/r04/teamab_$n.err
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r04/teamab_$n.json", errors="replace") if l.startswith("{")][-1])
print("$n", "code ms", round(d["kernel_ms"]["code"], 1), "ok", d["all_status_ok"], d["roofline"]["kernel_origin"])
PY
}
run default X=1
run wide ZPAQ_AMD_TEAM_ROWLANES=32
run prio1 ZPAQ_AMD_SPEC_DEFS=-DZPQ_TEAM_PRIO=1
run wide_prio2 ZPAQ_AMD_TEAM_ROWLANES=32 ZPAQ_AMD_SPEC_DEFS=-DZPQ_TEAM_PRIO=2
