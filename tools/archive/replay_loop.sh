#!/bin/bash
# diagnostics: is the single-rank replay ever wrong?  tiny-dims epoch vs oracle, N times
for rep in $(seq 1 ${1:-10}); do
  python -m pytest tests/test_hip_corpus_gpu.py -q -m gpu -k epoch 2>&1 | tail -1
done
