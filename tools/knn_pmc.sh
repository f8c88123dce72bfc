#!/bin/bash
# tools/knn_pmc.sh <tag> -- ON THE GPU BOX: SQ counters of round 3's and round 4's search kernels on the shapes of tools/knn_ab.py
# (two --pmc passes, no tracing flags: gpurun rules).  Needs tools/ab/libknn_ab.so (tools/build_variant.sh knn_ab -DMCP_AB).
set -e
tag=${1:-knnpmc}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp MCP_HIP_LIB=tools/ab/libknn_ab.so
timeout -k 5 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/p1" -o a -- python3 tools/knn_ab.py --once > "$out/p1.log" 2>&1
timeout -k 5 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/p2" -o b -- python3 tools/knn_ab.py --once > "$out/p2.log" 2>&1
python3 tools/knn_pmc_table.py $(find "$out/p1" "$out/p2" -name '*counter_collection.csv') > "$out/table.txt"
rm -rf "$out/p1" "$out/p2"
cat "$out/table.txt"
