#!/bin/bash
# Register / LDS / scratch usage of every kernel in one built object: tools/kernel_regs.sh change3d_amd/lib/obj/pw_gemm.o
set -e
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c '
import sys, re, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and "name" in cur and "vgpr_count" in cur: rows.append(cur); cur = {}
    if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count"):
        if k == "name" and "name" in cur and "vgpr_count" in cur: rows.append(cur); cur = {}
        if k == "name" and v.startswith("_Z") : cur["name"] = v
        elif k != "name": cur[k] = v
if "vgpr_count" in cur: rows.append(cur)
for r in rows:
    n = subprocess.run(["c++filt", r.get("name", "?")], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    print("v=%-4s a=%-3s s=%-4s scratch=%-5s spill=%-4s lds=%-6s %s" % (r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"), r.get("private_segment_fixed_size"), r.get("vgpr_spill_count"), r.get("group_segment_fixed_size"), n[:150]))
'
rm -rf $T
