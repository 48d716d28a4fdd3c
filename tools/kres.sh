#!/bin/bash
# kernel resource usage of one HIP source (registers, spills, LDS): tools/kres.sh distance.hip [filter]
src=slam.net_amd/csrc/$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --cuda-device-only -c $src -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | paste - - - - - - - - | sed 's/remark: [^ ]* //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/  */ /g' | grep "${2:-.}"
