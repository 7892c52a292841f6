#!/bin/bash
# correctness + timing of the implicit-GEMM convolution mode of csrc/gemm_ws.h on the decoder's shapes (GPU box)
S="conv 8 112 112 128 128 3 3 0 0  8 56 56 128 128 3 3 0 1  8 112 112 32 128 3 3 1 0  8 28 28 768 256 3 3 0 0  8 56 56 256 128 7 1 0 0  8 56 56 128 128 1 7 0 1  8 28 28 256 128 7 1 0 0  2 9 11 32 36 3 3 1 1  1 7 7 64 64 7 1 0 0  3 5 2 96 128 1 7 0 1"
for sp in 0 1; do echo "== split $sp"; WS_SPLIT=$sp timeout -k 10 150 ./tools/micro/gemm_ws_bench $S | grep -v "^AMD\|^,"; done
