#!/bin/bash
# build_variants.sh tag1="-DFLAGS" tag2="..." : tuning builds of the engine into build/var/libsmg_<tag>.so (parallel)
cd "$(dirname "$0")/../smudgeplot_amd/csrc"
mkdir -p ../../build/var
for spec in "$@"; do
  tag="${spec%%=*}"; flags="${spec#*=}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -I../../include $flags -shared \
      -o ../../build/var/libsmg_$tag.so smg_hetmers.hip -lpthread -ldl 2> ../../build/var/$tag.log || echo "BUILD FAILED $tag" ) &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done
wait
ls -la ../../build/var/*.so
