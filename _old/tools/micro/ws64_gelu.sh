#!/bin/bash
S="1568 1536 384 1 0 6272 768 192 1 0 25088 384 96 1 0 392 3072 768 1 0 1960 3072 768 1 0 1568 1152 384 0 0"
echo "== two per CU"; WS_TILE=64 timeout -k 10 100 ./tools/micro/gemm_ws_bench $S | grep cli
echo "== one per CU"; WS_TILE=64 WS64_ONE=1 timeout -k 10 100 ./tools/micro/gemm_ws_bench $S | grep cli
