P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("abi_calls_per_step"))'
for u in 0 1 0 1; do echo "union=$u b64"; NNR_CNE_UNION=$u python bench.py --no_cpu_baseline 2>/dev/null | python -c "$P"; done
for u in 0 1; do echo "union=$u b8"; NNR_CNE_UNION=$u python bench.py --no_cpu_baseline --batch_size 8 --steps 40 2>/dev/null | python -c "$P"; done
for u in 0 1; do echo "union=$u b16"; NNR_CNE_UNION=$u python bench.py --no_cpu_baseline --batch_size 16 --steps 40 2>/dev/null | python -c "$P"; done
