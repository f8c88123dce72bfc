#!/bin/bash
# ON THE GPU BOX: tools/fps_under_load.py fusion with one variant library: run_one.sh <name without .so>
out=gpurun_out/fps_ab && mkdir -p $out
export MCP_HIP_LIB=$PWD/tools/ab/$1.so
echo "== $1" | tee -a $out/summary.txt
timeout -k 10 240 python3 tools/fps_under_load.py fusion 2>&1 | tee -a $out/summary.txt
