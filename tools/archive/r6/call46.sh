#!/bin/bash
# round 6, GPU call 46: stream / hardware-queue patterns at batch 64 (is the natural order the best one?), four interleaved rounds
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
rm -f gpurun_out/r06Q_ab.txt
ab() {
  echo -n "$2 | $1 : " >> gpurun_out/r06Q_ab.txt
  env $1 timeout 300 python bench.py $2 --no_cpu_baseline --no_secondary --no_isolated --steps 40 --sustained_seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'])" >> gpurun_out/r06Q_ab.txt 2>&1
}
for i in 1 2 3 4; do
  for p in "" "0,0,0,1" "1,0,0,1" "0,0,1" "0,1" "1" "4"; do
    ab "NNR_STREAM_BURN=$p" "--batch_size 64"
  done
done
sort gpurun_out/r06Q_ab.txt
