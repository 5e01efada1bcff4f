#!/bin/bash
O=gpurun_out/${1:-r04bd}; mkdir -p $O
timeout 150 python scripts/r04/config_models_only.py din_taobao_b8192 > $O/din_only.txt 2>&1; grep "b8192" $O/din_only.txt | head -1 | cut -c1-300; tail -2 $O/din_only.txt | cut -c1-200
