#!/bin/bash
# round 6, GPU call 52: the same with the C-ABI RCCL binding (the exchange recorded inside the launch tape)
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
for b in 8 64; do for x in "0 0" "1 1" "1 0"; do set -- $x; NNR_DP_NATIVE=$2 timeout 300 python tools/dp_step_timing.py --batch_size $b --exchange $1 2>&1 | grep "^batch" | sed 's/table_bucket_rule.*binding/binding/' | cut -c1-300; done; done
