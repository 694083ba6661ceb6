#!/bin/bash
# Where the CLI's wall time goes (GPU box): `kpeg` on a 64x64 file against the 8K file, and the raw file read / write times.
# Round 2: 0.29-0.42 s for the tiny file, 0.29-0.36 s for the 8K one -- process start-up and HIP initialisation are the floor.
cd "$(dirname "$0")/.."
python3 - <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
import bench
d = tempfile.mkdtemp()
open(d + '/img8k.jpg', 'wb').write(bench.synth_jpeg(7680, 4320))
open(d + '/tiny.jpg', 'wb').write(bench.synth_jpeg(64, 64))
cli = os.path.abspath('libkpeg_amd/kpeg')
for name in ['tiny.jpg', 'img8k.jpg', 'tiny.jpg', 'img8k.jpg', 'img8k.jpg']:
    t0 = time.perf_counter()
    subprocess.run([cli, name], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print(name, "%.3f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); raw = open(d + '/img8k.jpg','rb').read(); print("read jpg %.4f" % (time.perf_counter()-t0))
buf = bytes(99532899)
t0 = time.perf_counter(); open(d + '/x.ppm','wb').write(buf); print("write 99.5 MB %.4f" % (time.perf_counter()-t0))
PY
