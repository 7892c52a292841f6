#!/bin/bash
# build the GEMM harness and print the kernels' register use (run from anywhere)
cd "$(dirname "$0")" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function -save-temps=obj -Rpass-analysis=kernel-resource-usage "$@" gemm_ws_bench.hip -o gemm_ws_bench 2>&1 | grep -i "error\| VGPRs:\|VGPRs Spill\|scratch" | grep -v "gemm_ws_bench.hip" | sed 's/.*remark: *//' | sort | uniq -c
rm -f *.bc *.hipi *.hipfb *.out *-host-* *.resolution.txt *.o
