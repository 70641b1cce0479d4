#!/bin/bash
# register / spill counts of the kernels in one object file of the build (usage: tools/kernel_regs.sh hmx_tile_bf [name filter])
O=${1:-hmx_tile_bf}; F=${2:-k_tile}
T=$(mktemp -d); cd "$T" || exit 1
objcopy -O binary --only-section=.hip_fatbin /root/repo/harmony_amd/lib/obj/$O.o fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes k.co 2>/dev/null | grep -E "\.name:|\.vgpr_count|vgpr_spill|agpr_count|private_segment_fixed" | paste - - - - - | grep "$F" | sed 's/ \+/ /g' | (c++filt 2>/dev/null || cat)
rm -rf "$T"
