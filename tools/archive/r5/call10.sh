#!/bin/bash
# round 5, GPU call 10: bf16x3 weight-gradient (TN) micro-benchmark; NNR_BX3=1 after the cross-stream fix (dp + headline + tape tests, suite), interleaved default / bx3 A/B
mkdir -p gpurun_out
rm -f gpurun_out/r05l_*.txt
for shape in "400 400 112640 46" "200 400 112640 46" "400 200 112640 46" "300 1664 112640 24" "400 400 28160 24" "900 900 4352 8"; do
  timeout 300 tools/micro/bf16x3_tn $shape 2>&1 | tee -a gpurun_out/r05l_bf16x3_tn.txt
done
(NNR_BX3=1 timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -40) > gpurun_out/r05l_suite_bx3.log
tail -4 gpurun_out/r05l_suite_bx3.log
ab() {
  echo "$1 $2" >> gpurun_out/r05l_ab.txt
  env $1 timeout 300 python bench.py --no_cpu_baseline --no_secondary --no_isolated --steps 40 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['value'], {k: (v['ms'], v['tflops'], v['launches']) for k, v in d['roofline']['families'].items() if 'bx3' in k})" >> gpurun_out/r05l_ab.txt 2>&1
}
for i in 1 2 3; do
  ab "NNR_BX3=0" ""
  ab "NNR_BX3=1" ""
done
ab "NNR_BX3=0" "--config mhsa"
ab "NNR_BX3=1" "--config mhsa"
cat gpurun_out/r05l_ab.txt
timeout 600 python bench.py --no_cpu_baseline --no_isolated --steps 40 --secondary_steps 10 > gpurun_out/r05l_bench.json 2> gpurun_out/r05l_bench.err
python -c "import json; d=json.loads(open('gpurun_out/r05l_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['secondary'].get('experimental_bf16x3_nt_cne_sue_b64'), indent=1))"
