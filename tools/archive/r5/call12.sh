#!/bin/bash
# round 5, GPU call 12: TN bf16x3 micro-benchmark v3 with / without the bank swizzle + SQ counters of both
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/r5/call11.sh "${1:-10 13 14}" > /dev/null
cp gpurun_out/r05m_bf16x3_tn.txt gpurun_out/r05n_bf16x3_tn.txt
ROOT=$PWD
cd /tmp
for v in ${2:-10 13}; do
  rm -rf /tmp/pmc_tn_$v
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_tn_$v -- $ROOT/tools/micro/bf16x3_tn 400 400 112640 32 $v > /tmp/pmc_tn_$v.log 2>&1
  python3 - $v <<'PY' >> $ROOT/gpurun_out/r05n_bf16x3_tn.txt
import csv, glob, sys, collections
v = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/tmp/pmc_tn_%s/**/*counter_collection.csv' % v, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
for k, d in acc.items():
    if 'bx3' in k: print('PMC variant', v, k, 'launches', cnt[k], {c: round(x / max(1, cnt[k])) for c, x in sorted(d.items())})
PY
done
cat $ROOT/gpurun_out/r05n_bf16x3_tn.txt | cut -c1-400
