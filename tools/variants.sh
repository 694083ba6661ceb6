#!/bin/bash
# Build experiment variants of libkpeg_hip.so into build/ablate/ (git-ignored):
#   tools/variants.sh NAME="-DX=1 -DY" NAME2="..."     then on the GPU box: tools/ablate_full.sh
set -e
cd "$(dirname "$0")/.."
rm -rf build/ablate; mkdir -p build/ablate; cp build/keep/*.so build/ablate/ 2>/dev/null || true   # reference builds kept from earlier trees
n=0
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}; [ "$defs" = "$spec" ] && defs=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -Iinclude $defs \
      -o build/ablate/libkpeg_hip_$name.so libkpeg_amd/csrc/kpeg_hip.hip 2>build/ablate/$name.log &
  n=$((n+1)); [ $((n % 6)) = 0 ] && wait
done
wait
ls build/ablate/*.so
