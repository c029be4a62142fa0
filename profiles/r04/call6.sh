#!/bin/bash
# GPU call 6 of round 4: the drop-in through the UNMODIFIED reference archiver after the submission queue learnt to wait for
# callers that arrive milliseconds apart: add / extract -method 50 over 256 x 1 MiB files, threads 16 / 64 / 256.
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
(time timeout 560 python profiles/r04/cli_bench.py --files 256 --timeout 150 --out gpurun_out/r04/cli.json) > gpurun_out/r04/cli.log 2>&1
grep -a '^{' gpurun_out/r04/cli.log
