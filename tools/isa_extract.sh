#!/bin/bash
# gfx950 disassembly of one built object: tools/isa_extract.sh change3d_amd/lib/obj/pw_gemm.o out_dir
#   -> out_dir/k.s (all kernels, llvm-objdump -d) ; then  python tools/isa_flow.py out_dir/k.s "<kernel name substring>"
set -e
mkdir -p "$2"
objcopy -O binary --only-section=.hip_fatbin "$1" "$2/fat.bin"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$2/fat.bin" --output="$2/k.co" --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn "$2/k.co" > "$2/k.s"
grep -n "^[0-9a-f]* <" "$2/k.s" | c++filt | sed 's/(anonymous namespace):://g' | cut -c1-160
