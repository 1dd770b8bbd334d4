#!/bin/bash
# Resource table (registers, spills, scratch, LDS) and code size of every kernel of the product library, read from the
# gfx950 code objects of the per-unit build (build/libfqsx/*.o).  usage: tools/kernel_resources.sh [build dir]
B=${1:-$(dirname $0)/../build/libfqsx}
L=/opt/rocm/lib/llvm/bin
for o in $B/fqsx_*.o; do
  objcopy -O binary --only-section=.hip_fatbin $o /tmp/kr_$$.fb 2>/dev/null; [ -s /tmp/kr_$$.fb ] || continue
  $L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/kr_$$.fb --output=/tmp/kr_$$.elf --unbundle 2>/dev/null || continue
  [ -s /tmp/kr_$$.elf ] || continue
  echo "== $(basename $o): .text $($L/llvm-readelf -S /tmp/kr_$$.elf | awk '$3==".text"{print $7}' | python3 -c 'import sys; print(int(sys.stdin.read().strip() or "0", 16))') bytes"
  $L/llvm-readelf --notes /tmp/kr_$$.elf | grep -E "\.name:|sgpr_count|vgpr_count|agpr_count|spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - - - | sed 's/  */ /g; s/\t/ /g' |
    awk '{for(i=1;i<=NF;i++){if($i==".name:")n=$(i+1); if($i==".vgpr_count:")v=$(i+1); if($i==".agpr_count:")a=$(i+1); if($i==".sgpr_count:")s=$(i+1); if($i==".sgpr_spill_count:")ss=$(i+1); if($i==".vgpr_spill_count:")vs=$(i+1); if($i==".private_segment_fixed_size:")p=$(i+1); if($i==".group_segment_fixed_size:")g=$(i+1)} printf "%-22s vgpr %3d agpr %3d sgpr %3d  sgpr_spill %4d vgpr_spill %3d scratch %4d B  lds %6d B\n",n,v,a,s,ss,vs,p,g}'
  # function sizes (roles are functions of their own)
  $L/llvm-readelf -s /tmp/kr_$$.elf 2>/dev/null | python3 -c '
import sys
rows = []
for l in sys.stdin:
    f = l.split()
    if len(f) >= 8 and f[3] == "FUNC":
        rows.append((int(f[2]), f[7]))
for n, name in sorted(rows)[-12:]:
    print("   %8d B  %s" % (n, name))'
done
rm -f /tmp/kr_$$.elf /tmp/kr_$$.fb
