#!/bin/bash
# Register / scratch / occupancy summary of every kernel in csrc/pt_kernels.hip (cross-compiled, no GPU needed). Extra flags: EXTRA=-D...
cd "$(dirname "$0")/../thu-acg-f2024-path-tracer_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $EXTRA -c csrc/pt_kernels.hip -o /tmp/pt_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: //' -e 's/ \[-Rpass.*//' | paste - - - - - \
 | sed -e 's/Function Name: _ZN2pt//' -e 's/    */ /g' | cut -c1-200
