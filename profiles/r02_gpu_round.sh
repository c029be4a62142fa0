#!/bin/bash
# One GPU call: the -m gpu suite, the headline bench (BASELINE configs[2]), configs[1], the dense batch and the decoder.
#   gpurun --timeout 2400 -- 'bash profiles/r02_gpu_round.sh'
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r02/gputest.txt
python bench.py > gpurun_out/r02/bench_headline.json 2> gpurun_out/r02/bench_headline.err
python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --steps 3 > gpurun_out/r02/bench_configs1.json 2> gpurun_out/r02/bench_configs1.err
python bench.py --blocks 2048 --cpu-seconds 0 --api-blocks 0 > gpurun_out/r02/bench_dense.json 2> gpurun_out/r02/bench_dense.err
python bench.py --mode decode --cpu-seconds 0 --api-blocks 0 > gpurun_out/r02/bench_decode.json 2> gpurun_out/r02/bench_decode.err
tail -3 gpurun_out/r02/gputest.txt
for f in headline configs1 dense decode; do python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r02/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", "value=%.1f MB/s" % j["value"], "code_ms=%.1f" % j["kernel_ms"]["code"], "frac=%.4f" % j["roofline"]["frac"], "ok=", j["all_status_ok"],
          "api=", (j.get("api") or {}).get("value"), "cpu=", (j.get("cpu_baseline") or {}).get("value"), "identical=", (j.get("cpu_baseline") or {}).get("bit_identical_vs_reference"),
          (j.get("cpu_baseline") or {}).get("compared_how"))
except Exception as e:
    print("$f FAILED", e, open("gpurun_out/r02/bench_$f.err").read()[-600:])
PY
done
