#!/bin/bash
# tools/grad_pmc.sh <tag> -- ON THE GPU BOX: SQ counters of the backward kernels inside one training run (tools/train_step_time.py),
# two --pmc passes, no tracing flags (gpurun rules).
set -e
tag=${1:-gradpmc}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/p1" -o a -- python3 tools/train_step_time.py 8 8192 eval > "$out/p1.log" 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/p2" -o b -- python3 tools/train_step_time.py 8 8192 eval > "$out/p2.log" 2>&1
python3 tools/pmc_kernel_table.py fusion_grad_kernel,cross_grad_kernel,attention_dkv_kernel,attention_dq_kernel,ptblock_grad_kernel,pointconv_agg_grad_kernel,fusion_split_kernel \
    $(find "$out/p1" "$out/p2" -name '*counter_collection.csv') > "$out/table.txt"
rm -rf "$out/p1" "$out/p2"
cat "$out/table.txt"
