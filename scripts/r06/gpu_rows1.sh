#!/bin/bash
# gpurun -- bash scripts/r06/gpu_rows1.sh <tag>: parity of the tall-input Linear kernels on the GPU + their timing against the library
tag=${1:-r06m}; out=gpurun_out/$tag; mkdir -p $out
export PYTORCH_TUNABLEOP_ENABLED=${PYTORCH_TUNABLEOP_ENABLED:-0}
timeout 900 python -m pytest tests/test_gemm_rows.py -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/gpu_tests.log
timeout 900 python scripts/r06/rows_gemm_bench.py > $out/rows_gemm_bench.txt 2>&1; echo "bench rc=$?"; cat $out/rows_gemm_bench.txt
