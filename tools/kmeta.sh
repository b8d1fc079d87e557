#!/bin/bash
# kernel metadata (VGPRs, SGPRs, spills, LDS, scratch) of the built library's gfx950 code objects: `bash tools/kmeta.sh [regex]`
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d /tmp/kmeta.XXXX); PAT=${1:-.}; LL=/opt/rocm/lib/llvm/bin
for o in $R/dgcnn_amd/csrc/*.o; do
  b=$(basename $o .o)
  $LL/llvm-objcopy --dump-section .hip_fatbin=$T/$b.fat $o 2>/dev/null || continue
  (cd $T && $LL/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$b.fat --output=$b.co --unbundle 2>/dev/null) || continue
  $LL/llvm-readelf --notes $T/$b.co 2>/dev/null | PAT="$PAT" python3 -c "
import sys,re,os
t=sys.stdin.read(); pat=os.environ['PAT']
for blk in re.split(r'\n\s*- \.agpr_count', t)[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)', blk) or [None,'?'])[1]
    n=g('name')
    if re.search(pat, n): print(f'{n[:72]:72s} vgpr {g(\"vgpr_count\"):>4s} sgpr {g(\"sgpr_count\"):>4s} spill {g(\"vgpr_spill_count\"):>3s} lds {g(\"group_segment_fixed_size\"):>6s} scratch {g(\"private_segment_fixed_size\"):>5s}')
"
done
rm -rf $T
