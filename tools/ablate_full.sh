#!/bin/bash
# run every build/ablate/libkpeg_hip_*.so through the full bench (timing experiments)
cd "$(dirname "$0")/.."
for f in build/ablate/libkpeg_hip_*.so; do
  v=$(basename $f .so); v=${v#libkpeg_hip_}
  KPEG_HIP_LIB=$PWD/$f python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$v', d['value'], d['ms_per_step'], 'unstuff %.3f sync %.3f scan %.3f write %.3f idct %.3f' % (k['unstuff_ms'],k['huff_sync_ms'],k['huff_scan_ms'],k['huff_write_ms'],k['idct_ms']), 'passes', d['sync_passes'])"
done
