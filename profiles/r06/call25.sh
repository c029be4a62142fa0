#!/bin/bash
# GPU call 25 of round 6: counters on the final code object (WRITE_SIZE, FETCH_SIZE up to three attempts: which passes return is
# not a property of the kernel); the LDS-rich latency shape for -m5 again, now that the ROW halves' re-fetch no longer waits
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
ZPAQ_AMD_LOG=1 ZPAQ_AMD_PERSIST_TIMEOUT_MS=20000 timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c25_pmc_persist_WRITE_SIZE -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c25_pmc_persist_WRITE_SIZE.log 2>&1
echo "persist WRITE_SIZE rc=$?"
for k in 1 2 3; do
  ZPAQ_AMD_LOG=1 ZPAQ_AMD_PERSIST_TIMEOUT_MS=20000 timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c25_pmc_persist_FETCH_SIZE_$k -o p -- python $R/profiles/pmc_driver.py 1024 1048576 98304 > $O/c25_pmc_persist_FETCH_SIZE_$k.log 2>&1
  rc=$?; echo "persist FETCH_SIZE attempt $k rc=$rc"; [ $rc = 0 ] && break
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/c25_pmc_*/")):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k, v in sorted(tot.items()):
        print(d.split("/")[-2], k[0], k[1], "dispatches=%d" % cnt[k], "sum=%.1f" % v)
PY
tail -3 $O/c25_pmc_persist_WRITE_SIZE.log
find $O -name "*.db" -delete 2>/dev/null
run() { name=$1; shift; env "$@" timeout 400 python bench.py $B > $O/c25_$name.json 2> $O/c25_$name.err; }
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 64";  run m5_64_def X=1;  run m5_64_rich ZPAQ_AMD_SMALL_CHAIN_WAVES=400
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 256"; run m5_256_def X=1; run m5_256_rich ZPAQ_AMD_SMALL_CHAIN_WAVES=400
B="--cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 512"; run m5_512_def X=1; run m5_512_rich ZPAQ_AMD_SMALL_CHAIN_WAVES=400
ZPAQ_AMD_SMALL_CHAIN_WAVES=400 ZPAQ_AMD_PERSIST_PROF=$O/c25_prof_m5_64_rich.bin timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --configs1 0 --legacy 0 --api-blocks 0 --verify-blocks 16 --blocks 64 --warmup 0 > /dev/null 2>&1
python profiles/persist_prof.py $O/c25_prof_m5_64_rich.bin > $O/c25_prof_m5_64_rich.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c25_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(j["value"], 1), "MB/s ok", j["all_status_ok"], "verified", j["roundtrip_verified_blocks"], "persist", j["persistent_launch"], "code ms", round(j["kernel_ms"]["code"], 1),
              (j.get("reference_identity") or {}).get("identical"), j["roofline"]["kernel_origin"][:20])
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
