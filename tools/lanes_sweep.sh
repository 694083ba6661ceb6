#!/bin/bash
cd "$(dirname "$0")/.."
for v in L2 L3 L4 L6 L8; do
  for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q KPEG_HIP_LIB=$PWD/build/ablate/libkpeg_hip_$v.so python bench.py --batch 256 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v hwq$q', d['value'], d['us_per_image'], d['one_stream']['us_per_image'])"
  done
done
