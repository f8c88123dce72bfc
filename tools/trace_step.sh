#!/bin/bash
# tools/trace_step.sh <tag> [bench args] -- ON THE GPU BOX: kernel trace of one bench run -> gpurun_out/<tag>/{kernel_stats.csv,step_by_queue.txt,step_timeline.txt}
set -e
tag=${1:-trace}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > "$out/trace_bench.json" 2> "$out/trace.err"
db=$(find "$out/trace" -name '*.db' | head -1)
python3 tools/rocpd_stats.py "$db" "$out/kernel_stats.csv" "$out/step_by_queue.txt" > /dev/null
python3 tools/step_timeline.py "$db" > "$out/step_timeline.txt" 2>&1 || true
python3 tools/step_timeline.py "$db" --all > "$out/step_timeline_all.txt" 2>&1 || true
rm -rf "$out/trace"
