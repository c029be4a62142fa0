#!/bin/bash
# GPU call 23 of round 6: the LDS-rich latency shape (ZPAQ_AMD_SMALL_CHAIN_WAVES) beyond the small chains -- -m5 on 64 / 256 / 512
# blocks with every latency-shape chain in it (400) against the default (32); -m4 on 256 / 1024 blocks and mid.cfg with the
# threshold at 32 (default) against 16; the small-chain GPU test and the encoder tests on the fixed initialisation
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "small_chains or method_3 or legacy or both_shapes or nine_component or random_hcomp or encode_matches" > $O/c23_tests.txt 2>&1
tail -5 $O/c23_tests.txt
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c23_$name.json 2> $O/c23_$name.err; }
B="$B --blocks 64";  run m5_64_def X=1;  run m5_64_rich ZPAQ_AMD_SMALL_CHAIN_WAVES=400
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 256"; run m5_256_def X=1; run m5_256_rich ZPAQ_AMD_SMALL_CHAIN_WAVES=400
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 512"; run m5_512_def X=1; run m5_512_rich ZPAQ_AMD_SMALL_CHAIN_WAVES=400
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --method 4 --blocks 256"; run m4_256_32 X=1; run m4_256_16 ZPAQ_AMD_SMALL_CHAIN_WAVES=16
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --method 4 --blocks 1024"; run m4_1024_32 X=1; run m4_1024_16 ZPAQ_AMD_SMALL_CHAIN_WAVES=16
B="--legacy-level 2 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 0 --api-blocks 0"; run legacy2_32 X=1; run legacy2_16 ZPAQ_AMD_SMALL_CHAIN_WAVES=16
ZPAQ_AMD_SMALL_CHAIN_WAVES=400 ZPAQ_AMD_PERSIST_PROF=$O/c23_prof_m5_64_rich.bin timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 64 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c23_prof_m5_64_rich.bin > $O/c23_prof_m5_64_rich.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c23_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
