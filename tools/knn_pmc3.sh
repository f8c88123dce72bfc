#!/bin/bash
# instruction-fetch counters of the two search kernels (one --pmc pass; see tools/knn_pmc.sh)
set -e
out=gpurun_out/${1:-knnpmc3}
mkdir -p "$out"
export TMPDIR=/tmp MCP_HIP_LIB=tools/ab/libknn_ab.so
timeout -k 5 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/p3" -o c -- python3 tools/knn_ab.py --once > "$out/p3.log" 2>&1
python3 tools/knn_pmc_table.py $(find "$out/p3" -name '*counter_collection.csv') > "$out/table.txt"
rm -rf "$out/p3"
