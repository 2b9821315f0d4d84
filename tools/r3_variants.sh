#!/bin/bash
# GPU-box helper: time the model kernels of every variants/libvad_<glob>.so.  usage: tools/r3_variants.sh 'p*' "model mode" ...
pat=$1; shift
for m in "$@"; do
  for lib in variants/libvad_$pat.so; do
    VAD_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/bench_model.py $m 50 2>/dev/null || exit 1
  done
done
