#!/bin/bash
# Build an experimental copy of the library: tools/build_variant.sh NAME SRC_DIR [extra hipcc flags ...]
# -> variants/libvad_NAME.so, selected at run time with VAD_AMD_LIB (tools/bench_fe.py, tools/bench_model.py).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xclang -target-feature -Xclang -packed-fp32-ops \
    -I"$root/include" -I"$src" "$@" -o "$root/variants/libvad_$name.so" \
    "$src"/{abi,mfcc,rnn,densenet,nicknet,nicknet_stream,overlay,sweep,seq}.hip 2>&1 | grep -v "packed-fp32-ops" || true
ls -la "$root/variants/libvad_$name.so"
