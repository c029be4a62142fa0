#!/bin/bash
# What does the unmodified archiver hand the library?  48 files of 1 MiB, -threads 16, one log line per device batch.
W=/tmp/zpq_probe; rm -rf $W; mkdir -p $W/tree
python - <<PY
import sys; sys.path.insert(0, '.')
from zpaq_amd import corpus
for i in range(48):
    corpus.block("text", 1 << 20, 777 + i).tofile(f"$W/tree/f{i:03d}.txt")
PY
cd $W
export GPU_MAX_HW_QUEUES=8 ZPAQ_AMD_LOG=1
( time timeout 170 $GRAFT_REPO_ROOT/oracle/_ref/zpaq_amd_cli add ours.zpaq tree -method 50 -threads 16 ) > add.log 2>&1
grep -F "[zpaq_amd]" add.log | head -60; grep -E "seconds|real" add.log
