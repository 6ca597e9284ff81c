#!/bin/bash
# build_r06.sh tag1="-DFLAGS" ... : tuning builds of the engine into build/r06/libsmg_<tag>.so, built HERE (they travel to the
# GPU box with the tree: build/r06 is git-ignored, not gpurun-ignored)
cd "$(dirname "$0")/../smudgeplot_amd/csrc"
mkdir -p ../../build/r06
for spec in "$@"; do
  tag="${spec%%=*}"; flags="${spec#*=}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -I../../include $flags -shared \
      -o ../../build/r06/libsmg_$tag.so smg_hetmers.hip -lpthread -ldl 2> ../../build/r06/$tag.log || echo "BUILD FAILED $tag" ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
ls -la ../../build/r06/*.so
