#!/bin/bash
# Round 3, GPU call 2: are the six pipe streams falsely serialised by sharing 4 hardware queues (GPU_MAX_HW_QUEUES)?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c2
mkdir -p $O
cd $R
T0=$(date +%s)
for q in 8 16; do
  echo "== A/B headline, GPU_MAX_HW_QUEUES=$q" | tee -a $O/summary.txt
  GPU_MAX_HW_QUEUES=$q timeout 600 python profiles/ab_inproc.py profiles/r03/ab2_queues.json --out $O/ab2_q$q.jsonl > $O/ab2_q$q.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import json
for ln in open("$O/ab2_q$q.jsonl"):
    j = json.loads(ln)
    if "error" in j: print("%-18s ERROR %s" % (j["name"], j["error"][:160])); continue
    print("%-18s %8.1f ms %7.1f MB/s ok=%s same=%s" % (j["name"], j["code_ms"], j["MBps"], j["status_ok"], j["same_bytes_as_first"]))
PY
  echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
done
B64="python bench.py --blocks 1024 --block-bytes 65536 --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1"
ALL="ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1"
for v in default all; do
  E=""; [ $v = all ] && E="$ALL"
  env $E GPU_MAX_HW_QUEUES=8 ZPAQ_AMD_SPEC_DEFS=-DZPQ_TRACE ZPAQ_AMD_PIPE_TRACE=$O/trace_$v.bin timeout 300 $B64 > $O/trace_$v.json 2> $O/trace_$v.err
  echo "== placement $v (8 hardware queues)"; python profiles/pipe_trace.py $O/trace_$v.bin 2>&1 | head -24
  python - <<PY
import numpy as np
a = np.fromfile("$O/trace_$v.bin", dtype=np.uint64).reshape(-1, 4)
np.savez_compressed("$O/trace_$v.npz", a=a[:min(len(a), 400000)])
PY
  rm -f $O/trace_$v.bin
done 2>&1 | tee -a $O/summary.txt
echo "t=$(( $(date +%s) - T0 )) s" | tee -a $O/summary.txt
