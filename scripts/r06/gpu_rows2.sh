#!/bin/bash
# gpurun -- bash scripts/r06/gpu_rows2.sh <tag>: the tall-input Linear kernels, product library and experiment variants, one box
tag=${1:-r06q}; out=gpurun_out/$tag; mkdir -p $out
export PYTORCH_TUNABLEOP_ENABLED=0
timeout 600 python scripts/r06/rows_gemm_bench.py > $out/rows_gemm_bench.txt 2>&1; echo "bench rc=$?"
for v in ${VARIANTS:-nothing nobar nothingnobar}; do
  ROWS_LIB=libtzrec_hip_$v.so ROWS_NO_CHECK=1 timeout 600 python scripts/r06/rows_gemm_bench.py > $out/rows_gemm_bench_$v.txt 2>&1; echo "$v rc=$?"
done
for f in $out/rows_gemm_bench*.txt; do echo "== $f"; grep -v amdgpu.ids $f | cut -c1-100; done
