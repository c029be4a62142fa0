#!/bin/bash
# GPU call 16 of round 4: the lockstep decoder's phase profile by bit position (a byte's first bit, the second nibble's first bit, the rest).
set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
export ZPAQ_AMD_SPEC_DEFS=-DZPQ_PROF
(time timeout 200 python bench.py --mode decode --kernel 6 --blocks 2048 --block-bytes 1048576 --cpu-seconds 0 --warmup 0 --steps 1 --verify-blocks 0) > gpurun_out/r04/dec_pos_prof.json 2> gpurun_out/r04/dec_pos_prof.err
grep -a "team prof" gpurun_out/r04/dec_pos_prof.json gpurun_out/r04/dec_pos_prof.err
