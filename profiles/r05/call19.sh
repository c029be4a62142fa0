#!/bin/bash
# GPU call 19 of round 5: the coder with buffered output (four coded bytes per store, lanes flushing together once per
# input byte): identity against the oracle and the step kernels, configs[1] with the per-unit profile, the headline, small batches
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 400 python profiles/r05/persist_check.py 60 > $O/c19_check.log 2>&1; tail -6 $O/c19_check.log
ZPAQ_AMD_PERSIST_PROF=$O/c19_prof_configs1.bin timeout 300 python bench.py --method 3 --kind lcg --blocks 256 --block-bytes 262144 --decode-blocks 0 --cpu-seconds 0 --api-blocks 0 > $O/c19_configs1.json 2> $O/c19_configs1.err
python -c "import json; d=json.load(open('$O/c19_configs1.json')); print('configs1', round(d['value'],1), d['persistent_launch'], d['kernel_ms'], d.get('reference_identity'))"
python profiles/persist_prof.py $O/c19_prof_configs1.bin | head -12
timeout 300 python bench.py --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c19_headline.json 2> $O/c19_headline.err
python -c "import json; d=json.load(open('$O/c19_headline.json')); print('headline', round(d['value'],1), d['persistent_launch'], d['kernel_ms'], d.get('reference_identity'))"
for nb in 64 256; do
  ZPAQ_AMD_PERSIST_PROF=$O/c19_prof_$nb.bin timeout 200 python bench.py --blocks $nb --cpu-seconds 0 --decode-blocks 0 --api-blocks 0 > $O/c19_b$nb.json 2> $O/c19_b$nb.err
  python -c "import json; d=json.load(open('$O/c19_b$nb.json')); print($nb, round(d['value'],1), d['persistent_launch'], d['kernel_ms'])"
done
python profiles/persist_prof.py $O/c19_prof_64.bin | head -30
