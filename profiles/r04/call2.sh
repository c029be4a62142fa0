#!/bin/bash
# GPU call 2 of round 4: the lockstep decoder's first run on the MI355X -- parity (golden archives, odd batch, 2048-block
# launch against the other decoders), the configs[4] operating point (2048 x 1 MiB) with it, and a probe of what the
# reference archiver hands the library (call 1: `add -method 50 -threads 16` crawled at 1.4 MB/s).
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
(time timeout 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lockstep or cross_lane") > gpurun_out/r04/call2_pytest.txt 2>&1
tail -15 gpurun_out/r04/call2_pytest.txt
(time timeout 300 python bench.py --mode decode --kernel 6 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/bench_decode_team.json 2> gpurun_out/r04/bench_decode_team.err
tail -c 1500 gpurun_out/r04/bench_decode_team.json; tail -5 gpurun_out/r04/bench_decode_team.err
bash profiles/r04/cli_probe.sh > gpurun_out/r04/cli_probe.txt 2>&1
tail -70 gpurun_out/r04/cli_probe.txt
