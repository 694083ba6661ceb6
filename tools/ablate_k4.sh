#!/bin/bash
# Build timing-only variants of libkpeg_hip.so (kernel experiments; outputs are wrong by design)
# into gpurun_out-independent dir build/ablate/, to be run on the GPU box with tools/ablate_run.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ablate
for v in NONE BARRIERS COLOUR STORES LOADS "$@"; do
  defs=""
  for d in ${v//+/ }; do [ "$d" != NONE ] && defs="$defs -DKPEG_ABLATE_$d"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $defs \
      -o build/ablate/libkpeg_hip_$v.so libkpeg_amd/csrc/kpeg_hip.hip &
done
wait
ls build/ablate
