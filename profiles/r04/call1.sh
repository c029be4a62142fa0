#!/bin/bash
# GPU call 1 of round 4: the round-3 code with the new bench line (decode leg beside the headline) and the drop-in
# measured through the unmodified reference archiver.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
(time python bench.py) > gpurun_out/r04/bench_base.json 2> gpurun_out/r04/bench_base.err
tail -c 3000 gpurun_out/r04/bench_base.json
(time python profiles/r04/cli_bench.py --out gpurun_out/r04/cli.json) > gpurun_out/r04/cli.log 2>&1
tail -30 gpurun_out/r04/cli.log
