#!/bin/bash
# tools/profile_round.sh <tag> -- run ON THE GPU BOX (gpurun -- bash tools/profile_round.sh r02_a): the kernel trace of one bench
# run plus the three PMC passes tools/pmc_counters.py merges.  Writes gpurun_out/<tag>/; copy the summaries into profiles/.
# rocprofv3 gets the program itself after `--` (python3 ...), and --pmc passes carry no tracing flags (gpurun rules).
set -e
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o t -- $B > "$out/trace_bench.json" 2> "$out/trace.err"
db=$(find "$out/trace" -name '*.db' | head -1)
python3 tools/rocpd_stats.py "$db" "$out/kernel_stats.csv" "$out/step_by_queue.txt" > /dev/null
echo "trace done"
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o f -- $B > "$out/pmc_fetch.log" 2>&1
echo "fetch done"
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o w -- $B > "$out/pmc_write.log" 2>&1
echo "write done"
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/pmc_sq" -o s -- $B > "$out/pmc_sq.log" 2>&1
echo "sq done"
python3 -c "import json, bench; json.dump({'csrc_sha256': bench.kernel_sources_digest()}, open('$out/pmc_sources.json', 'w'))"
python3 tools/pmc_counters.py "$out/pmc.json" $(find "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq" -name '*counter_collection.csv') | tee "$out/pmc_summary.txt"
# the raw per-dispatch CSVs are large: keep only the merged JSON and the stats
python3 tools/step_timeline.py "$db" > "$out/step_timeline.txt" 2>&1 || true
python3 tools/step_timeline.py "$db" --all > "$out/step_timeline_all.txt" 2>&1 || true
rm -rf "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq" "$out/trace"
