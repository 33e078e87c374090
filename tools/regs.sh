#!/bin/bash
# register / scratch / occupancy summary of one kernel source: tools/regs.sh rows_gemm
cd /root/repo/point_dae_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage -c $1.hip -o /tmp/regs_$1.o 2>&1 | grep -E "error|Function Name|  VGPRs:|AGPRs|Occupancy|SGPRs Spill|ScratchSize" | paste - - - - - - | sed -E 's/.*Function Name: //; s/\[-Rpass[^]]*\]//g; s/[a-z_]+.hip:[0-9]+:[0-9]+: remark://g; s/ +/ /g; s/\[bytes\/lane\]//; s/\[waves\/SIMD\]//'
