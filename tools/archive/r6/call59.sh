#!/bin/bash
# round 6, GPU call 59: soaks at the final commit: 5 000 replayed steps at batch 64, 4 000 at batch 8
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
timeout 900 python3 tools/replay_soak.py --steps 5000 > gpurun_out/r06_final_soak_b64.json 2> gpurun_out/r06_final_soak_b64.err
timeout 900 python3 tools/replay_soak.py --steps 4000 --batch_size 8 > gpurun_out/r06_final_soak_b8.json 2> gpurun_out/r06_final_soak_b8.err
for f in b64 b8; do python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r06_final_soak_$f.json') if l.startswith('{')][-1]); print('$f', {k: d[k] for k in d if k != 'loss_every_100_steps'})"; done
