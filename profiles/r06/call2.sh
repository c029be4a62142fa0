#!/bin/bash
# GPU call 2 of round 6: call 1 says the packed MIX rows made the launch SLOWER (the MIX wavefronts are the longest chain of the
# launch: 619 -> 811 / 1209 instructions per byte) -- so is the launch bound by the MIX wavefronts' own stream?  (a) MIX split in
# halves (two lane groups per block, half the bits each) on the round-5 units, with the per-unit profile of both;
# (b) the lockstep decoder with the tail wavefront against round 5's form: parity tests + the decode leg
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 64"
export ZPAQ_AMD_MIX_PACKED=0
ZPAQ_AMD_MIX_HALVES=1 timeout 400 python bench.py $B > $O/c2_head_halves.json 2> $O/c2_head_halves.err
ZPAQ_AMD_MIX_HALVES=1 ZPAQ_AMD_PERSIST_PROF=$O/c2_prof_halves.bin timeout 300 python bench.py $B --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c2_prof_halves.bin > $O/c2_prof_halves.txt 2>&1
ZPAQ_AMD_PERSIST_PROF=$O/c2_prof_unpacked.bin timeout 300 python bench.py $B --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c2_prof_unpacked.bin > $O/c2_prof_unpacked.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lockstep or decode_reference or decoder_status or foreign_kernel" -s > $O/c2_tests.txt 2>&1
tail -8 $O/c2_tests.txt
D="--cpu-seconds 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 0 --decode-blocks 2048"
timeout 900 python bench.py $D > $O/c2_dec_tail.json 2> $O/c2_dec_tail.err
ZPAQ_AMD_TEAM_TAIL=0 timeout 900 python bench.py $D > $O/c2_dec_notail.json 2> $O/c2_dec_notail.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c2_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        d = j.get("decode") or {}
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), "| decode", d.get("value"), d.get("every_byte_verified"), (d.get("ms") or {}).get("code"), (d.get("roofline") or {}).get("kernel_origin"), d.get("error"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -60 $O/c2_prof_halves.txt
echo ---- unpacked
head -60 $O/c2_prof_unpacked.txt
