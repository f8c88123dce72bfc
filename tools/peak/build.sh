#!/bin/bash
# builds the matrix-pipe measurement helper into build/libmfma_peak.so (gfx950)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o build/libmfma_peak.so tools/peak/mfma_peak.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o build/libsync_latency.so tools/peak/sync_latency.hip
