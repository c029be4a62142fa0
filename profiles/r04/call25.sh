#!/bin/bash
# GPU call 25 of round 4: the LZ77 parse and the BWT on the device (device/lz77_kernel.h): parity tests, then configs[1] with
# the parse on the device and (knob) on the host.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
(time timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lz77_parse or suffix_arrays or method_3_known or preprocessing") > gpurun_out/r04/gputest_lz77.txt 2>&1
tail -15 gpurun_out/r04/gputest_lz77.txt
for knob in 1 0; do
  export ZPAQ_AMD_DEVICE_PARSE=$knob
  (time timeout 200 python bench.py --method 3 --blocks 256 --block-bytes 262144 --kind lcg --decode-blocks 0 --steps 3 --cpu-seconds 0) > gpurun_out/r04/bench_configs1_parse$knob.json 2> gpurun_out/r04/bench_configs1_parse$knob.err
  python - $knob <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/r04/bench_configs1_parse%s.json" % sys.argv[1], errors="replace") if l.startswith("{")][-1])
print("device parse =", sys.argv[1], "value", round(d["value"], 1), "api", round(d["api"]["value"], 1), d["api"]["ms"])
PY
done
export ZPAQ_AMD_DEVICE_PARSE=1
(time timeout 200 python bench.py --method 3 --blocks 256 --block-bytes 262144 --kind text --decode-blocks 0 --steps 3 --cpu-seconds 0) > gpurun_out/r04/bench_m3_text_parse1.json 2> gpurun_out/r04/bench_m3_text_parse1.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_m3_text_parse1.json", errors="replace") if l.startswith("{")][-1])
print("text, device parse: value", round(d["value"], 1), "api", round(d["api"]["value"], 1), d["api"]["ms"])
PY
