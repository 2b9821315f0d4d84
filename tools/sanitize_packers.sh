#!/bin/bash
# Host-side AddressSanitizer + UBSan run of every weight packer (vad_pack_net_weights for hidden 1..96, both heads,
# GRU and LSTM; vad_pack_densenet_weights; vad_pack_nicknet_weights): tests/c_client/pack_sanitize.cpp calls them
# with random tensors of the reference's shapes.  No GPU is needed (the packers are host code); the device code is
# compiled unsanitized.  Takes about 90 s.   usage: tools/sanitize_packers.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/vad_pack_sanitize
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
    -Xarch_host -fno-omit-frame-pointer -Xclang -target-feature -Xclang -packed-fp32-ops \
    -I"$root/include" -I"$root/voice_activity_detection_amd/csrc" -o "$out" "$root/tests/c_client/pack_sanitize.cpp" \
    "$root"/voice_activity_detection_amd/csrc/{abi,mfcc,rnn,densenet,nicknet,nicknet_stream,overlay,sweep,seq}.hip 2>&1 | grep -v "packed-fp32-ops" || true
"$out"
