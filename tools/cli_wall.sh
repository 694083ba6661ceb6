#!/bin/bash
# Wall time of the drop-in CLI on the 8K headline image (GPU box): `kpeg <file.jpg>` = process + HIP start-up, parse, decode,
# download, 99.5 MB PPM written.  usage: tools/cli_wall.sh [runs]
cd "$(dirname "$0")/.."
python3 - "${1:-5}" <<'PY'
import hashlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
import bench
d = tempfile.mkdtemp()
open(d + '/img8k.jpg', 'wb').write(bench.synth_jpeg(7680, 4320))
cli = os.path.abspath('libkpeg_amd/kpeg')
for i in range(int(sys.argv[1])):
    if os.path.exists(d + '/img8k.ppm'):
        os.remove(d + '/img8k.ppm')
    t0 = time.perf_counter()
    subprocess.run([cli, 'img8k.jpg'], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    raw = open(d + '/img8k.ppm', 'rb').read()
    print("kpeg img8k.jpg: %.3f s wall, ppm %d bytes, sha256 %s" % (dt, len(raw), hashlib.sha256(raw).hexdigest()[:16]), flush=True)
PY
