#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...] -- a second build of the library (A/B or diagnostic defines) as
# tools/ab/lib<name>.so, objects under build/<name>/; the product library (mocopci_amd/libmocopci_hip.so) is not touched.
# Select it at run time with MCP_HIP_LIB=tools/ab/lib<name>.so (mocopci_amd/_lib.py).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/mocopci_amd/csrc
out=$root/build/$name
mkdir -p "$out" "$root/tools/ab"
flags=$(sed -n 's/^FLAGS   ?= //p' "$src/Makefile" | sed 's/\$(ARCH)/gfx950/; s/\$(MFMA_FORM)//')
# the per-file flags of the Makefile: the register form of MFMA results everywhere but fusion_grad / ptblock_grad, no NaN canonicalisation
# in the fusion / vector-attention files, no atomic optimizer in fps
ls "$src"/*.hip | xargs -P 8 -I{} bash -c 'f={}; b=$(basename $f .hip); extra="-mllvm -amdgpu-mfma-vgpr-form"; case $b in fusion_grad|ptblock_grad) extra="";; esac; case $b in fusion|ptblock|fusion_grad|ptblock_grad) extra="$extra -fno-honor-nans";; esac; [ $b = fps ] && extra="$extra -mllvm -amdgpu-atomic-optimizer-strategy=None"; /opt/rocm/bin/hipcc '"$flags $*"' $extra -c $f -o '"$out"'/$b.o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab/lib$name.so" "$out"/*.o
echo "built tools/ab/lib$name.so"
