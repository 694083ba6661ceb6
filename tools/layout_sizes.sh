#!/bin/bash
# Whole-decode time by image size and coefficient layout, variants interleaved in one process (GPU box).
#   tools/variants.sh cur="" ; tools/layout_sizes.sh [variant ...]
cd "$(dirname "$0")/.."
for sz in "1920 1080" "3840 2160" "7680 4320"; do set -- $sz "${@:3}"; echo "== $1 x $2"; python tools/k4_ab.py --mode decode --layouts 1,2 --width $1 --height $2 ${VARIANTS:-cur} 2>&1 | grep -v amdgpu.ids; done
