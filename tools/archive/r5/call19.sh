#!/bin/bash
# round 5, final build: full GPU suite, then every judged artefact (tools/r5/collect.sh)
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -30) > gpurun_out/r05z_suite.log
tail -5 gpurun_out/r05z_suite.log
bash tools/r5/collect.sh r05z > gpurun_out/r05z_collect.log 2>&1
tail -30 gpurun_out/r05z_collect.log
