#!/usr/bin/env python3
"""tests/golden/make_golden_large.py -- pin the BASELINE-sized inputs to the REAL reference.

Build container only (needs oracle/_ref/kpeg_ref, i.e. /root/reference; Pillow and scikit-learn's
bundled sample photographs for the natural-content fixtures):

    python tests/golden/make_golden_large.py [--skip-16k]

Writes tests/golden/manifest_large.json:
  synth   BASELINE configs 2-4: the seed-1234.. synthetic 1920x1080 files (32 seeds = the distinct
          images of the 256-image batch) and the 7680x4320 file, decoded by libKPEG's own decoder;
          only SHA-256s are kept (the generator is integer-only, the GPU box regenerates the inputs
          byte for byte and checks jpg_sha256 first).
  dri16k  BASELINE config 5: 16384x16384 with one restart interval per MCU row.  The reference
          rejects DRI, so every interval is re-wrapped as a standalone 16384x8 JFIF file
          (SOI + the original APP0/DQT/SOF0 with the height patched to 8/DHT + SOS + interval
          bytes + EOI) and decoded by the reference in a process of its own;
          SHA-256 of the whole PPM and of the raw RGB bytes of each of 8 row stripes.
  natural Pillow encodings (4:4:4, 2-6 bits per pixel) of photographs: inputs committed
          (tests/golden/nat_*.jpg), PPM hashes from the reference.
Every reference output is also compared with oracle/kpeg_oracle.c here: the restatement is
thereby pinned at the full sizes, not only on the small fixtures.
"""
import concurrent.futures as cf
import hashlib
import io
import json
import os
import struct
import subprocess
import sys
import tempfile
import shutil
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kpeg_testlib as T  # noqa: E402

OUT = os.path.join(HERE, "manifest_large.json")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def ref_decode_file(path):
    out = subprocess.run([T.REF_BIN, "decode", path], capture_output=True, text=True, timeout=7200)
    return json.loads(out.stdout.strip().splitlines()[-1])


def ref_ppm(data, tag):
    """(info, ppm bytes) of the real reference on `data`, one fresh process."""
    d = tempfile.mkdtemp(prefix="kpeggl_" + tag)
    try:
        f = os.path.join(d, "in.jpg")
        open(f, "wb").write(data)
        info = ref_decode_file(f)
        assert info["status"] == "DECODE_DONE", (tag, info)
        return info, open(os.path.join(d, "in.ppm"), "rb").read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pin_synth(w, h, seed):
    data = T.synth_jpeg(w, h, seed=seed)
    t0 = time.time()
    info, ppm = ref_ppm(data, "s%d" % seed)
    st, want = T.oracle_decode(data, nthreads=1)
    assert st == T.DECODE_DONE and T.ppm_bytes(want) == ppm, "oracle differs from the reference on synth %dx%d seed %d" % (w, h, seed)
    return {"width": w, "height": h, "seed": seed, "quality": 75, "sigma": 6.0, "jpg_sha256": sha(data), "jpg_bytes": len(data),
            "ppm_sha256": sha(ppm), "rgb_sha256": sha(ppm[ppm.index(b"\n255\n") + 5:]), "ref_decode_s": round(info["decode_s"], 2),
            "wall_s": round(time.time() - t0, 1)}


def split_header(data):
    """(bytes before DRI, bytes after DRI up to and including the SOS header, scan, tail)."""
    i = data.find(b"\xff\xdd\x00\x04")
    assert i > 0
    sos = data.find(b"\xff\xda")
    sos_len = struct.unpack(">H", data[sos + 2:sos + 4])[0]
    scan0 = sos + 2 + sos_len
    assert data[-2:] == b"\xff\xd9"
    return data[:i], data[i + 6:scan0], data[scan0:-2]


def patch_height(hdr, h):
    j = hdr.find(b"\xff\xc0")
    assert j > 0
    return hdr[:j + 5] + struct.pack(">H", h) + hdr[j + 7:]


def _decode_interval(args):
    k, blob = args
    info, ppm = ref_ppm(blob, "i%d" % k)
    return k, ppm[ppm.index(b"\n255\n") + 5:]


def pin_dri(w, h, seed, workers):
    mw = w // 8
    t0 = time.time()
    data = T.synth_jpeg(w, h, seed=seed, restart_interval=mw)
    print("  generated %d bytes in %.0f s" % (len(data), time.time() - t0), flush=True)
    pre, post, scan = split_header(data)
    # restart markers: FF D0..D7 cannot occur inside entropy-coded data (FF is always followed by 00 there)
    a = np.frombuffer(scan, np.uint8)
    ff = np.flatnonzero((a[:-1] == 0xFF) & (a[1:] >= 0xD0) & (a[1:] <= 0xD7))
    nint = h // 8
    assert len(ff) == nint - 1, (len(ff), nint)
    bounds = [0] + [int(x) + 2 for x in ff]
    ends = [int(x) for x in ff] + [len(scan)]
    hdr = patch_height(pre + post, 8)
    jobs = ((k, hdr + scan[bounds[k]:ends[k]] + b"\xff\xd9") for k in range(nint))
    rows = [None] * nint
    with cf.ThreadPoolExecutor(workers) as ex:   # threads: each job is a subprocess of its own
        for n, (k, rgb) in enumerate(ex.map(_decode_interval, jobs)):
            assert len(rgb) == w * 8 * 3
            rows[k] = rgb
            if n % 256 == 0:
                print("  interval %d / %d  (%.0f s)" % (n, nint, time.time() - t0), flush=True)
    full = b"".join(rows)
    del rows
    want, _, _ = T.oracle_decode_rst(data, mw, 8)
    assert want.tobytes() == full, "oracle differs from the per-interval reference decode"
    import ctypes
    L = T.oracle()
    buf = ctypes.create_string_buffer(256)
    n = L.kpeg_oracle_ppm_header(w, h, buf, 256)
    hdr_ppm = buf.raw[:n]
    hh = hashlib.sha256()
    hh.update(hdr_ppm)
    hh.update(full)
    stripe = h * w * 3 // 8
    return {"width": w, "height": h, "seed": seed, "quality": 75, "sigma": 6.0, "restart_interval": mw, "jpg_sha256": sha(data),
            "jpg_bytes": len(data), "ppm_sha256": hh.hexdigest(), "rgb_sha256": sha(full),
            "stripe8_rgb_sha256": [sha(full[s * stripe:(s + 1) * stripe]) for s in range(8)],
            "how": "each restart interval re-wrapped as a 16384x8 JFIF and decoded by the reference in its own process",
            "wall_s": round(time.time() - t0, 1)}


def pil_jpeg(rgb, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", **kw)
    return b.getvalue()


def pin_natural():
    from sklearn.datasets import load_sample_image
    out = {}
    china = load_sample_image("china.jpg")[:424, :640]
    flower = load_sample_image("flower.jpg")[:424, :640]
    cases = [("nat_china_640x424_q50", china, dict(quality=50)), ("nat_china_640x424_q90", china, dict(quality=90)),
             ("nat_flower_640x424_q75_opt", flower, dict(quality=75, optimize=True)),
             ("nat_flower_320x208_q96", flower[::2, ::2][:208, :320], dict(quality=96))]
    for name, rgb, kw in cases:
        rgb = np.ascontiguousarray(rgb)
        data = pil_jpeg(rgb, subsampling=0, **kw)
        info, ppm = ref_ppm(data, name)
        st, want = T.oracle_decode(data, nthreads=1)
        assert st == T.DECODE_DONE and T.ppm_bytes(want) == ppm, name
        open(os.path.join(HERE, name + ".jpg"), "wb").write(data)
        h, w = rgb.shape[:2]
        out[name] = {"width": w, "height": h, "jpg_sha256": sha(data), "jpg_bytes": len(data), "ppm_sha256": sha(ppm),
                     "bits_per_pixel": round(len(data) * 8 / (w * h), 2), "source": "scikit-learn sample photograph, Pillow %s" % kw}
        print("  %-30s %6d bytes  %.2f bit/px" % (name, len(data), out[name]["bits_per_pixel"]), flush=True)
    return out


def main():
    assert T.have_ref(), "oracle/_ref/kpeg_ref is missing: run `make -C oracle ref` in the build container"
    manifest = json.load(open(OUT)) if os.path.exists(OUT) else {}
    what = set(sys.argv[1:]) or {"natural", "1080p", "8k", "16k"}
    if "natural" in what:
        print("natural-content fixtures", flush=True)
        manifest["natural"] = pin_natural()
        json.dump(manifest, open(OUT, "w"), indent=1, sort_keys=True)
    if "1080p" in what:
        print("1080p x 32 seeds", flush=True)
        with cf.ThreadPoolExecutor(8) as ex:
            res = list(ex.map(lambda s: pin_synth(1920, 1080, s), range(1234, 1234 + 32)))
        manifest.setdefault("synth", {})
        for r in res:
            manifest["synth"]["1920x1080_seed%d" % r["seed"]] = r
        json.dump(manifest, open(OUT, "w"), indent=1, sort_keys=True)
    if "8k" in what:
        print("8K", flush=True)
        manifest.setdefault("synth", {})["7680x4320_seed1234"] = pin_synth(7680, 4320, 1234)
        json.dump(manifest, open(OUT, "w"), indent=1, sort_keys=True)
    if "16k" in what:
        print("16384x16384 DRI", flush=True)
        manifest["dri16k"] = pin_dri(16384, 16384, 1234, 8)
        json.dump(manifest, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
