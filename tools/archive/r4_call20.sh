#!/bin/bash
# Round 4 A/B: skinny GEMM with 8 / 16 waves and two groups in flight (NNR_SKINNY_WAVES=4: the round 1-3 split)
O=gpurun_out/r04q; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_ops_gpu.py -x -q -m gpu -k "gemm or skinny" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python3 bench.py --prebuilt --no_cpu_baseline --no_isolated --sustained_seconds 2"
for r in 1 2; do
$B > $O/bench_new_$r.json 2>> $O/err
NNR_SKINNY_WAVES=4 $B > $O/bench_w4_$r.json 2>> $O/err
$B --batch_size 8 --steps 40 --warmup 8 > $O/bench_b8_new_$r.json 2>> $O/err
NNR_SKINNY_WAVES=4 $B --batch_size 8 --steps 40 --warmup 8 > $O/bench_b8_w4_$r.json 2>> $O/err
done
NNR_SKINNY_WAVES=8 $B --batch_size 8 --steps 40 --warmup 8 > $O/bench_b8_w8_1.json 2>> $O/err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); fam=d['roofline']['families']
        sk=sum(v['ms'] for k,v in fam.items() if 'skinny' in k); n=sum(v['launches'] for k,v in fam.items() if 'skinny' in k)
        print('%-16s %8.1f %7.3f sustained %s  skinny %.1f us avg (%d sampled)' % (f.split('bench_')[1][:-5], d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), 1000*sk/max(n,1), n))
    except Exception as e: print(f, 'FAILED', e)
PY
