#!/bin/bash
# ON THE GPU BOX: tools/fps_under_load.py fusion, once per variant built by build_fps_variants.sh (and once for the shipped library)
out=gpurun_out/fps_ab && mkdir -p $out
for v in shipped libfps_old_packed libfps_old_packed_nop libfps_new_packed libfps_old_scalar; do
  if [ $v = shipped ]; then unset MCP_HIP_LIB; else export MCP_HIP_LIB=$PWD/tools/ab/$v.so; fi
  echo "== $v" | tee -a $out/summary.txt
  timeout -k 10 240 python3 tools/fps_under_load.py fusion 2>&1 | tee -a $out/summary.txt || exit 1
done
