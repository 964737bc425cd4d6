#!/bin/bash
# round 5, GPU call 4: split gradient norm (NNR_SPLIT_NORM), resident-ring soak test, full GPU suite
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/r05f_tests.log
tail -4 gpurun_out/r05f_tests.log
ab() {
  echo "$1 $2" >> gpurun_out/r05f_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], {k: v['avg_launch_us'] for k, v in d['roofline']['hbm'].items() if k in ('sumsq', 'clip_adam')})" >> gpurun_out/r05f_ab.txt
}
for i in 1 2; do
  ab "NNR_SPLIT_NORM=0" ""
  ab "NNR_SPLIT_NORM=1" ""
  ab "NNR_SPLIT_NORM=0" "--batch_size 8"
  ab "NNR_SPLIT_NORM=1" "--batch_size 8"
done
cat gpurun_out/r05f_ab.txt
timeout 600 python tools/replay_soak.py --steps 2000 --batch_size 8 --busy 48 > gpurun_out/r05f_soak_busy.json 2>gpurun_out/r05f_soak_busy.err; tail -c 600 gpurun_out/r05f_soak_busy.json
