#!/bin/bash
# register / spill summary of gemm3.hip's kernels (scratch tool)
cd /root/repo/effocr_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c gemm3.hip -o /tmp/gemm3_test.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning:|Function Name|VGPRs:|ScratchSize|SGPRs Spill|VGPRs Spill" | paste - - - - - | sed 's/\[-Rpass[^]]*\]//g; s/gemm3.hip:[0-9]*:1: remark://g; s/_ZN6effocr12_GLOBAL__N_112gemm3_kernel//' | cut -c1-200 | head -${1:-6}
