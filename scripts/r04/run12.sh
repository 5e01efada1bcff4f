#!/bin/bash
mkdir -p gpurun_out/r04v
hipcc --offload-arch=gfx950 -O2 scripts/r04/probe_mfma_4b.hip -o /tmp/probe4b && timeout 60 /tmp/probe4b > gpurun_out/r04v/probe_mfma_4b.txt 2>&1
cat gpurun_out/r04v/probe_mfma_4b.txt | tail -45
