#!/bin/bash
# round 5, final tree: full GPU suite (after the test-file fix), smoke(), long soaks of the final build
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/r05y_suite.log
tail -4 gpurun_out/r05y_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r05y_smoke.log
cat gpurun_out/r05y_smoke.log
timeout 900 python3 tools/replay_soak.py --steps 6000 > gpurun_out/r05y_soak_b64.json 2> gpurun_out/r05y_soak.err
tail -c 300 gpurun_out/r05y_soak_b64.json
timeout 900 python3 tools/replay_soak.py --steps 8000 --batch_size 8 --busy 48 > gpurun_out/r05y_soak_b8_busy.json 2>> gpurun_out/r05y_soak.err
tail -c 300 gpurun_out/r05y_soak_b8_busy.json
