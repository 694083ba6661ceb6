#!/bin/bash
# run every build/ablate/libkpeg_hip_*.so through the K4-only bench (timing experiments)
cd "$(dirname "$0")/.."
for f in build/ablate/libkpeg_hip_*.so; do
  v=$(basename $f .so); v=${v#libkpeg_hip_}
  for m in ${MODES:-0}; do
  KPEG_HIP_LIB=$PWD/$f python bench.py --steps 20 --warmup 3 --no-cpu-baseline --idct-only --idct-mode $m 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'mode$m', d['roofline']['kernel_ms'], d['roofline']['frac'], d['exact_pixels_per_image'])"
  done
done
