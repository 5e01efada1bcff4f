#!/bin/bash
O=gpurun_out/${1:-r04aq}; mkdir -p $O
timeout 400 python scripts/r04/config_models_only.py > $O/config_models.txt 2>&1; grep "b8192" $O/config_models.txt | cut -c1-400; tail -3 $O/config_models.txt | cut -c1-300
