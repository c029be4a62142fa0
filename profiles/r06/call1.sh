#!/bin/bash
# GPU call 1 of round 6: (a) what a request that stays on-die is worth (gups2); (b) the new GPU tests (legacy models at BASELINE
# block sizes, foreign kernel beside the persistent launch, decompress(n) prefix) and the encoder's parity tests on the packed
# MIX rows; (c) headline A/B: packed rows + 128 LDS rows / 64 / packed only / unpacked (round 5's units); (d) legacy levels 2, 3
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 300 profiles/r06/gups2 > $O/c1_gups2.txt 2>&1; cat $O/c1_gups2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_api.py -m gpu -x -q -k "legacy_models_at_baseline or foreign_kernel or prefix_on_the_device or encode_matches_oracle_and_golden or both_shapes or persistent_launch_gives_up or all_nine" -s > $O/c1_tests.txt 2>&1
tail -15 $O/c1_tests.txt
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64"
timeout 400 python bench.py $B > $O/c1_head_lds128.json 2> $O/c1_head_lds128.err
ZPAQ_AMD_PERSIST_PROF=$O/c1_prof_lds128.bin timeout 300 python bench.py $B --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c1_prof_lds128.bin > $O/c1_prof_lds128.txt 2>&1
ZPAQ_AMD_MIX_LDS_ROWS=64 timeout 400 python bench.py $B > $O/c1_head_lds64.json 2> $O/c1_head_lds64.err
ZPAQ_AMD_MIX_LDS_ROWS=0 timeout 400 python bench.py $B > $O/c1_head_lds0.json 2> $O/c1_head_lds0.err
ZPAQ_AMD_MIX_PACKED=0 timeout 400 python bench.py $B > $O/c1_head_unpacked.json 2> $O/c1_head_unpacked.err
timeout 300 python bench.py --legacy-level 2 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 4 > $O/c1_legacy2.json 2> $O/c1_legacy2.err
timeout 500 python bench.py --legacy-level 3 --kind text --blocks 1024 --block-bytes 1048576 --decode-blocks 0 --configs1 0 --legacy 0 --cpu-seconds 6 > $O/c1_legacy3.json 2> $O/c1_legacy3.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c1_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"],
              "code ms", round(j["kernel_ms"]["code"], 1), "frac", round(j["roofline"]["frac"], 4), (j.get("reference_identity") or {}).get("identical"),
              (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("bit_identical_vs_reference"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -40 $O/c1_prof_lds128.txt
