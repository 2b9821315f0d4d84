#!/bin/bash
# Time every variants/libvad_*.so matching the given glob with a tool: tools/run_variants.sh 'abl_*' tools/bench_fe.py 50
pat=$1; shift
for lib in variants/libvad_$pat.so; do
    VAD_AMD_LIB=$PWD/$lib timeout -k 10 120 python "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
