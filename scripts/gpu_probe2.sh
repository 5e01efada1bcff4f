#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python scripts/gemm_probe.py default
TORCH_BLAS_PREFER_HIPBLASLT=0 timeout 120 python scripts/gemm_probe.py rocblas
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunable.csv WARM=3 timeout 500 python scripts/gemm_probe.py tunableop_tune 2>&1 | tail -2
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunable.csv timeout 120 python scripts/gemm_probe.py tunableop_replay 2>&1 | tail -1
cp /tmp/tunable*.csv gpurun_out/ 2>/dev/null; wc -l gpurun_out/tunable*.csv
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$OLDPWD/gpurun_out/pmc_$c" -o p --output-format csv -- python "$OLDPWD/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
done
cd "$OLDPWD"; find gpurun_out/pmc_FETCH_SIZE -type f | head
