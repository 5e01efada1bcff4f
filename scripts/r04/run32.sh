#!/bin/bash
O=gpurun_out/${1:-r04ar}; mkdir -p $O
timeout 200 python scripts/r04/din_sync_debug.py > $O/din_sync_debug.txt 2>&1; grep -v "amdgpu.ids" $O/din_sync_debug.txt | tail -60 | cut -c1-220
