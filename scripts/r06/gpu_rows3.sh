#!/bin/bash
# gpurun -- bash scripts/r06/gpu_rows3.sh <tag>: DIN towers on the library's own tall-input products: parity on the GPU, the kernels
# alone (product library + variants), the multi_tower_din step with them and with the GEMM library (same box)
tag=${1:-r06r}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gemm_rows.py tests/test_sequence_parity.py tests/test_reference_module_vectors.py tests/test_dense_glue.py -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/gpu_tests.log
PYTORCH_TUNABLEOP_ENABLED=0 timeout 600 python scripts/r06/rows_gemm_bench.py > $out/rows_gemm_bench.txt 2>&1; echo "bench rc=$?"
for v in ${VARIANTS:-}; do
  PYTORCH_TUNABLEOP_ENABLED=0 ROWS_LIB=libtzrec_hip_$v.so ROWS_NO_CHECK=1 timeout 600 python scripts/r06/rows_gemm_bench.py > $out/rows_gemm_bench_$v.txt 2>&1; echo "$v rc=$?"
done
for f in $out/rows_gemm_bench*.txt; do echo "== $f"; grep -v amdgpu.ids $f | cut -c1-100; done
TZR_OWN_ROWS_GEMM=1 timeout 600 python scripts/r05/din_step.py 30 jagged > $out/din_step_own.txt 2>&1; echo "din own rc=$?"; tail -1 $out/din_step_own.txt
TZR_OWN_ROWS_GEMM=0 timeout 600 python scripts/r05/din_step.py 30 jagged > $out/din_step_library.txt 2>&1; echo "din lib rc=$?"; tail -1 $out/din_step_library.txt
