#!/bin/bash
# GPU-box helper: wide-GRU parity tests, then model-kernel timings.  usage: tools/r3_step.sh "model mode" ...
set -e
timeout -k 10 200 python -u -m pytest tests/test_gpu_models.py -x -q --timeout 120 --timeout-method thread -k "${R3_TESTS:-wide_gru or other_hidden}" 2>&1 | tail -3
for m in "$@"; do
  timeout -k 10 120 python tools/bench_model.py $m 50 2>/dev/null
done
