#!/bin/bash
# correctness + timing of the 64x64-tile persistent kernel (csrc/gemm_ws64.h) on the mid-size shapes of the forward (GPU box)
S="1568 1536 384 1 0 1568 384 1536 0 1 1568 1152 384 0 0 1568 384 384 0 1 1960 768 768 0 1 1960 3072 768 1 0 6272 768 192 1 0 6272 192 768 0 1 25088 384 96 1 0 25088 96 384 0 1 392 3072 768 1 0 392 768 3072 0 1 392 2304 768 0 0 1000 200 96 1 1 37 64 128 0 1 7840 512 512 0 1"
echo "== 64x64 tiles"; WS_TILE=64 timeout -k 10 150 ./tools/micro/gemm_ws_bench $S | grep -v "^AMD\|^,"
