#!/bin/bash
# final form of the generalised interaction backward: gpu tests + numbers
mkdir -p gpurun_out/r03bt
timeout 600 python -m pytest tests/test_interaction_parity.py tests/test_interaction_top.py -q -m gpu > gpurun_out/r03bt/pytest_interaction.txt 2>&1
tail -3 gpurun_out/r03bt/pytest_interaction.txt
timeout 300 python scripts/bench_interaction_gen.py > gpurun_out/r03bt/bench_interaction_gen_final.txt 2>&1
IA_GEN_SMALL=1 timeout 300 python scripts/bench_interaction_gen.py >> gpurun_out/r03bt/bench_interaction_gen_final.txt 2>&1
timeout 200 python scripts/bench_interaction.py > gpurun_out/r03bt/bench_interaction.txt 2>&1
cat gpurun_out/r03bt/bench_interaction_gen_final.txt gpurun_out/r03bt/bench_interaction.txt
