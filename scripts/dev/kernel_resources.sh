#!/bin/bash
# per-kernel register / scratch / LDS / occupancy of a HIP source as the gfx950 backend reports it
# usage: scripts/dev/kernel_resources.sh qp_kernel.hip [extra flags]
HERE=$(cd $(dirname $0)/../../bluerov2_amd/csrc && pwd)
SRC=${1:-qp_kernel.hip}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I$HERE/../../include --cuda-device-only -c $HERE/$SRC -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' \
  | awk '/Name:/{if (line) print line; line=$0; next} {line=line" |"$0} END{print line}' | sed 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/; s/LDS Size \[bytes\/block\]/lds/'
