#!/bin/bash
# K4-only timing of every build/ablate/libkpeg_hip_*.so for a list of grid sizes (wavefronts per CU; 0 = the library's default)
#   WAVES="0 16 19 20 38" tools/k4_sweep.sh
cd "$(dirname "$0")/.."
for f in build/ablate/libkpeg_hip_*.so; do
  v=$(basename $f .so); v=${v#libkpeg_hip_}
  for w in ${WAVES:-0}; do
    if [ "$w" = 0 ]; then unset KPEG_K4_WAVES_PER_CU; else export KPEG_K4_WAVES_PER_CU=$w; fi
    KPEG_DEBUG=1 KPEG_HIP_LIB=$PWD/$f python bench.py --steps 20 --warmup 3 --no-cpu-baseline --idct-only 2>build/ablate/err.txt | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'waves/cu=$w', d['roofline']['kernel_ms'], d['roofline']['frac'], d['exact_pixels_per_image'])"
    grep -h "K4 wavefronts" build/ablate/err.txt | head -1
  done
done
