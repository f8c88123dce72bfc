#!/bin/bash
# tools/ab/build_fps_variants.sh -- run HERE (needs .git): four builds of the furthest-point-sampling translation unit around commit
# fc3ad47 ("no packed-fp32 instructions"), each linked with the current objects of every other unit, for ONE run of
# tools/fps_under_load.py per variant on the GPU box (MCP_HIP_LIB=tools/ab/<name>.so):
#   A  libfps_old_packed.so       fps.hip of fc3ad47^ as it was built then (ext-vector pairs -> v_pk_*_f32)
#   B  libfps_old_packed_nop.so   A with `s_nop 1` inserted after every v_pk_*_f32 in the device assembly
#   C  libfps_new_packed.so       today's fps.hip (indices buffered in LDS, raw min/max) with the pairs as ext-vectors again
#   D  libfps_old_scalar.so       fps.hip of fc3ad47^ with the pairs as the scalar struct mcp_f2 (no LDS index buffering)
# A vs B: is it a missing wait state after packed ops?   A vs D: was "scalar" the fix?   A vs C: was the LDS index buffer the fix?
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD
LL=/opt/rocm/lib/llvm/bin
W=/tmp/fps_ab && rm -rf $W && mkdir -p $W/pkg/csrc $W/include $W/pkg2/csrc
git show fc3ad47^:mocopci_amd/csrc/fps.hip > $W/pkg/csrc/fps.hip
for f in common.h topk.h mfma_split.h; do git show fc3ad47^:mocopci_amd/csrc/$f > $W/pkg/csrc/$f; done
git show fc3ad47^:include/mocopci_hip.h > $W/include/mocopci_hip.h
OLDFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function"
NEWFLAGS="$OLDFLAGS -fno-slp-vectorize -fno-vectorize"
OTHERS=$(ls mocopci_amd/csrc/*.o | grep -v '/fps.o')
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/$1 $2 $OTHERS; }
# A (+ temps for B)
(cd $W/pkg/csrc && mkdir st && cd st && /opt/rocm/bin/hipcc $OLDFLAGS -save-temps -c ../fps.hip -o fps_packed.o 2>/dev/null)
link libfps_old_packed.so $W/pkg/csrc/st/fps_packed.o
# B
(cd $W/pkg/csrc/st && python3 - <<'PY'
out = []
n = 0
for l in open('fps-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n'):
    out.append(l)
    t = l.strip().split()
    if t and t[0].startswith('v_pk_') and t[0].endswith('_f32'):
        out.append('\ts_nop 1')
        n += 1
open('fps_nop.s', 'w').write('\n'.join(out))
print(n, 'packed ops padded')
PY
 $LL/clang -cc1as -triple amdgcn-amd-amdhsa -filetype obj -main-file-name fps.hip -target-cpu gfx950 -mrelocation-model pic -o fps_nop_dev.o fps_nop.s
 $LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols -plugin-opt=mcpu=gfx950 -o fps_nop.out fps_nop_dev.o
 $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=fps_nop.out -output=fps_nop.hipfb
 $LL/llvm-objcopy --update-section .hip_fatbin=fps_nop.hipfb fps_packed.o fps_packed_nop.o)
link libfps_old_packed_nop.so $W/pkg/csrc/st/fps_packed_nop.o
# C: today's source, pairs as ext-vectors
mkdir -p $W/new/csrc $W/include2 && cp mocopci_amd/csrc/*.h $W/new/csrc/ && mkdir -p $W/new/../include && cp include/mocopci_hip.h $W/include/mocopci_hip.h.new
mkdir -p $W/n/m/csrc $W/n/include && cp mocopci_amd/csrc/*.h $W/n/m/csrc/ && cp include/mocopci_hip.h $W/n/include/
sed -e 's|^typedef mcp_f2 f2;.*|typedef float f2 __attribute__((ext_vector_type(2)));|' \
    -e 's|^__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return mcp_f2_fma(a, b, c); }|__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }|' \
    mocopci_amd/csrc/fps.hip > $W/n/m/csrc/fps.hip
(cd $W/n/m/csrc && /opt/rocm/bin/hipcc $OLDFLAGS -c fps.hip -o fps_new_packed.o 2>/dev/null)
link libfps_new_packed.so $W/n/m/csrc/fps_new_packed.o
# D: old source, scalar pairs (today's common.h supplies mcp_f2)
mkdir -p $W/d/m/csrc $W/d/include && cp mocopci_amd/csrc/*.h $W/d/m/csrc/ && cp include/mocopci_hip.h $W/d/include/
sed -e 's|^typedef float f2 __attribute__((ext_vector_type(2)));|typedef mcp_f2 f2;|' -e 's|__builtin_elementwise_fma|mcp_f2_fma|g' $W/pkg/csrc/fps.hip > $W/d/m/csrc/fps.hip
(cd $W/d/m/csrc && /opt/rocm/bin/hipcc $NEWFLAGS -c fps.hip -o fps_old_scalar.o 2>/dev/null)
link libfps_old_scalar.so $W/d/m/csrc/fps_old_scalar.o
for so in tools/ab/*.so; do
  n=$($LL/llvm-objdump -d --offloading $so >/dev/null 2>&1; echo); echo "built $so"
done
