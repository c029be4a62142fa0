#!/bin/bash
# GPU call 36 of round 6: the north-star sweep and the archiver timings on the final code (512 x 16 MiB blocks and the archiver's
# two chains of ~255 blocks each run the latency shape's variant 1, which changed in call 34)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 900 python profiles/r05/cli_bench.py --files 256 --out $O/c36_cli.json > $O/c36_cli.log 2>&1
cut -c1-220 $O/c36_cli.log | tail -12
timeout 1500 python profiles/sweep_north.py $O/c36_sweep_north.jsonl > $O/c36_sweep.log 2>&1
python - <<PY
import json
for ln in open("$O/c36_sweep_north.jsonl"):
    j = json.loads(ln)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.items() if k in ("block_bytes", "blocks", "kind", "MBps", "roofline_frac", "cpu_MBps", "ok", "decoded_back", "blocks_identical_to_reference", "error", "skipped")})
PY
