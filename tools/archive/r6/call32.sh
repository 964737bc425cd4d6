#!/bin/bash
# round 6, GPU call 32: recurrence tile-step on the BF16 pipe (tools/micro/lstm_bx3.hip; verdict item 1c)
mkdir -p gpurun_out
timeout 300 tools/micro/lstm_bx3 256 256 > gpurun_out/r06F_lstm_bx3.txt 2>&1
timeout 300 tools/micro/lstm_bx3 256 64 >> gpurun_out/r06F_lstm_bx3.txt 2>&1
cat gpurun_out/r06F_lstm_bx3.txt
