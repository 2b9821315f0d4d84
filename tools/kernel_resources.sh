#!/bin/bash
# Registers, spills, LDS and occupancy of every kernel in one source file, as the compiler reports them.
# usage: tools/kernel_resources.sh rnn.hip [name-filter]
here="$(cd "$(dirname "$0")/.." && pwd)"
src="$here/voice_activity_detection_amd/csrc/$1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c -o /dev/null \
    -Xclang -target-feature -Xclang -packed-fp32-ops -I"$here/include" -I"$here/voice_activity_detection_amd/csrc" \
    -Rpass-analysis=kernel-resource-usage "$src" 2>&1 | c++filt | sed "s/(anonymous namespace):://g; s/vad:://g" |
  awk -v pat="${2:-.}" '
    /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name); sub(/\(.*/,"",name); show = (name ~ pat); if (show) printf "\n%s:", name }
    show && /(VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|Occupancy|LDS Size)/ { line=$0; sub(/.*remark: [^ ]* +/,"",line); sub(/ \[-Rpass.*/,"",line); printf "  %s;", line }
    END { print "" }'
