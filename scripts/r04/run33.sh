#!/bin/bash
O=gpurun_out/${1:-r04as}; mkdir -p $O
timeout 200 python -m pytest tests/test_graph_pipeline_gpu.py -x -q -m gpu -k "sequence_model" > $O/test_din_capture.txt 2>&1; tail -5 $O/test_din_capture.txt
