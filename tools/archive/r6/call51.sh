#!/bin/bash
# round 6, GPU call 51: a rank's batch-8 / batch-64 step with the gradient exchange in it (one-rank RCCL communicator) vs without, same process layout
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
for b in 8 64; do for x in 0 1 0 1; do timeout 300 python tools/dp_step_timing.py --batch_size $b --exchange $x 2>&1 | grep "^batch" | cut -c1-400; done; done
