#!/usr/bin/env python3
"""tests/golden/make_golden_photos.py -- pin the 8K natural-content inputs of bench.py to the REAL reference.

Build container only (needs oracle/_ref/kpeg_ref, i.e. /root/reference, and Pillow):

    python tests/golden/make_golden_photos.py

For every case of bench.PHOTO_CASES (a committed photograph tiled to 7680 x 4352 and re-encoded by the integer-only test
encoder) libKPEG's own decoder writes the PPM; the SHA-256 of the JPEG bytes and of the raw RGB bytes go to
tests/golden/manifest_large.json under "natural_8k" (the GPU box regenerates the inputs and checks jpg_sha256 first).  The
oracle (oracle/kpeg_oracle.c) must give the same pixels: the restatement is pinned on natural content at full size too.
"""
import concurrent.futures as cf
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import shutil
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import kpeg_testlib as T  # noqa: E402

OUT = os.path.join(HERE, "manifest_large.json")


def one(case):
    src, q = case
    data = bench.tiled_photo_jpeg(src, q)
    d = tempfile.mkdtemp(prefix="kpegph")
    try:
        f = os.path.join(d, "in.jpg")
        open(f, "wb").write(data)
        t0 = time.time()
        out = subprocess.run([T.REF_BIN, "decode", f], capture_output=True, text=True, timeout=7200)
        info = json.loads(out.stdout.strip().splitlines()[-1])
        assert info["status"] == "DECODE_DONE", info
        raw = open(os.path.join(d, "in.ppm"), "rb").read()
        rgb = raw.split(b"\n", 4)[4]
        st, want = T.oracle_decode(data, 2)
        assert st == T.DECODE_DONE and want.tobytes() == rgb, "oracle != reference on %s q%d" % (src, q)
        p = T.oracle_parse(data)
        return "%s_q%d_%dx%d" % (os.path.splitext(src)[0], q, bench.PHOTO_W, bench.PHOTO_H), {
            "source": src, "quality": q, "width": bench.PHOTO_W, "height": bench.PHOTO_H, "jpg_bytes": len(data),
            "bits_per_pixel": round(len(p.scan) * 8 / (bench.PHOTO_W * bench.PHOTO_H), 3),
            "jpg_sha256": hashlib.sha256(data).hexdigest(), "rgb_sha256": hashlib.sha256(rgb).hexdigest(),
            "ref_decode_s": info.get("decode_s"), "wall_s": round(time.time() - t0, 1)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    assert T.have_ref(), "needs oracle/_ref/kpeg_ref (make -C oracle ref)"
    man = json.load(open(OUT))
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        res = dict(ex.map(one, bench.PHOTO_CASES))
    man["natural_8k"] = res
    json.dump(man, open(OUT, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print(k, v["bits_per_pixel"], "bits/px, reference took", v["ref_decode_s"], "s")


if __name__ == "__main__":
    main()
