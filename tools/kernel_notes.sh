#!/bin/bash
# notes.sh <lib.so> [kernel-substring]: per-kernel resource notes of the gfx950 code object inside a fat shared library
lib=$1; pat=${2:-kf_pass1_d}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$lib --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co.o 2>/dev/null || \
  { python3 - "$lib" "$tmp/co.o" <<'PY'
import sys
d=open(sys.argv[1],'rb').read()
i=d.find(b'__CLANG_OFFLOAD_BUNDLE__')
import struct
n=struct.unpack_from('<Q',d,i+24)[0]
o=i+32
for _ in range(n):
    off,size,tl=struct.unpack_from('<QQQ',d,o); t=d[o+24:o+24+tl]; o+=24+tl
    if b'gfx950' in t: open(sys.argv[2],'wb').write(d[i+off:i+off+size])
PY
  }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/co.o | awk -v pat="$pat" '
  /\.name:/ {name=$2}
  /\.sgpr_count:|\.sgpr_spill_count:|\.vgpr_count:|\.vgpr_spill_count:|\.group_segment_fixed_size:|\.private_segment_fixed_size:/ {v[$1]=$2}
  /\.wavefront_size:/ { if (name ~ pat) printf "%s sgpr %s spill %s vgpr %s spill %s lds %s scratch %s\n", name, v[".sgpr_count:"], v[".sgpr_spill_count:"], v[".vgpr_count:"], v[".vgpr_spill_count:"], v[".group_segment_fixed_size:"], v[".private_segment_fixed_size:"] }'
rm -rf $tmp
