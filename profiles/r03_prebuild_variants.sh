#!/bin/bash
# Run HERE (no GPU needed) before `gpurun ... profiles/r03_ab_bits.sh`: compiles the code objects of every knob combination
# that script measures into zpaq_amd/spec_cache/ (which travels to the GPU box), so that no GPU minute is spent in hipRTC.
# The knobs are part of the generated source and therefore of the cache key; ZPAQ_AMD_KEEP_CACHE keeps the other variants.
cd "$(dirname "$0")/.."
export ZPAQ_AMD_KEEP_CACHE=1
run() { echo "== $*"; env "$@" python -m zpaq_amd.prebuild; }
run ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1
for d in 1 2 3 4; do run ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_MIX_DEPTH=$d; done
for d in 1 2 3; do run ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=$d; done
for m in 1 2 4 7; do run ZPAQ_AMD_PIPE_LIGHT_BITS=$m; done
run ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_ROW_NIBBLES=1
run ZPAQ_AMD_PIPE_FULL_SQUASH=1
run ZPAQ_AMD_PIPE_ROW_FLAT=1
run ZPAQ_AMD_PIPE_MAP_ILP=2
run ZPAQ_AMD_PIPE_MAP_ILP=4
run ZPAQ_AMD_PIPE_MAP_ILP=2 ZPAQ_AMD_PIPE_FULL_SQUASH=1
run ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_MAP_ILP=2 ZPAQ_AMD_PIPE_FULL_SQUASH=1
run ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=3 ZPAQ_AMD_PIPE_FULL_SQUASH=1 ZPAQ_AMD_PIPE_MAP_ILP=2
run ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=4
run ZPAQ_AMD_PIPE_ROW_NIBBLES=1 ZPAQ_AMD_PIPE_ROW_DEPTH=3 ZPAQ_AMD_PIPE_FULL_SQUASH=1
for d in 2 3 4; do run ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_MIX_DEPTH=$d ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_LIGHT_DEPTH=$d ZPAQ_AMD_PIPE_ROW_NIBBLES=1; done
run ZPAQ_AMD_PIPE_CHUNK=1024
run ZPAQ_AMD_PIPE_CHUNK=1024 ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1
run ZPAQ_AMD_SPEC_DEFS=-DZPQ_TRACE
run ZPAQ_AMD_SPEC_DEFS=-DZPQ_TRACE ZPAQ_AMD_PIPE_MIX_BITS=1 ZPAQ_AMD_PIPE_LIGHT_BITS=7 ZPAQ_AMD_PIPE_ROW_NIBBLES=1
python -m zpaq_amd.prebuild          # the default set last (and first in the cache listing of the product)
ls zpaq_amd/spec_cache | wc -l
