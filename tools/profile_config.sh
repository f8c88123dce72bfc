#!/bin/bash
# tools/profile_config.sh <c4|c5> <tag> -- ON THE GPU BOX: kernel-trace stats + the FETCH_SIZE / WRITE_SIZE / SQ counter passes of
# `bench.py --config <c>` (BASELINE configs[3] / [4]); summaries under gpurun_out/<tag>/ (copy into profiles/).
set -e
cfg=$1; tag=${2:-prof_$1}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $B > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
cp $(find "$out/trace" -name '*kernel_stats.csv' | head -1) "$out/kernel_stats.csv"
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o f -- $B > "$out/pmc_fetch.log" 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o w -- $B > "$out/pmc_write.log" 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/pmc_sq" -o s -- $B > "$out/pmc_sq.log" 2>&1
python3 tools/pmc_counters.py "$out/pmc.json" $(find "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq" -name '*counter_collection.csv') | tee "$out/pmc_summary.txt"
rm -rf "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq" "$out/trace"
