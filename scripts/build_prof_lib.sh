#!/bin/bash
# libtzrec_hip_prof.so = the product library + -DIT_PROF (in-kernel phase clocks of the fused interaction kernels);
# built in-tree so that it travels to the GPU box; loaded only by scripts/bench_interaction_top.py --prof
cd "$(dirname "$0")/../torcheasyrec_amd" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -I csrc -DIT_PROF -shared csrc/*.hip -o libtzrec_hip_prof.so
