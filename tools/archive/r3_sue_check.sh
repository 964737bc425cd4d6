#!/bin/bash
# Round-3 SUE kernel changes (csrc/gcn.hip LDS footprint, csrc/misc.hip sue_intra_*): unit tests of the touched kernels, the SUE
# layer / model parity tests, then the per-call timeline of the replayed batch-64 step.   tools/r3_sue_check.sh TAG
TAG=${1:-r03x}
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops_gpu.py -q -m gpu -k "gcn or sue_intra" > gpurun_out/${TAG}_ops.log 2>&1
tail -3 gpurun_out/${TAG}_ops.log
python -m pytest tests/test_hip_layers_gpu.py tests/test_hip_model_gpu.py tests/test_hip_tape_gpu.py -q -m gpu -x > gpurun_out/${TAG}_layers.log 2>&1
tail -3 gpurun_out/${TAG}_layers.log
python tools/tape_timeline.py > gpurun_out/${TAG}_timeline_b64.txt 2>&1
head -1 gpurun_out/${TAG}_timeline_b64.txt
grep -E "sue_|gcn_" gpurun_out/${TAG}_timeline_b64.txt
python bench.py --no_cpu_baseline --sustained_seconds 2 > gpurun_out/${TAG}_bench.json 2>gpurun_out/${TAG}_bench.err
python -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench.json'));print(d['value'],d['ms_per_step'],d.get('sustained'))"
