#!/usr/bin/env python3
"""tests/golden/make_golden.py -- (re)generate the golden fixtures from the REAL reference.

Run in the build container only (needs oracle/_ref/kpeg_ref, i.e. /root/reference, and Pillow):

    python tests/golden/make_golden.py

The reference publishes no test vectors, so every fixture here is an
input/output pair produced by running the reference decoder itself (one fresh process per
image) on
  * its own sample misc/images/lena.jpg (only the SHA-256 of the PPM is kept; the input stays
    under /root/reference),
  * small synthetic JPEGs from tools/kpeg_synth.c and Pillow-encoded JPEGs (inputs committed),
  * malformed / unsupported files (only the decoder's ResultCode is recorded).
Outputs: tests/golden/*.jpg, *.ppm (small images) and manifest.json.
"""
import hashlib
import io
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kpeg_testlib as T  # noqa: E402


def sha(b):
    return hashlib.sha256(b).hexdigest()


def ref_status(data):
    import tempfile
    d = tempfile.mkdtemp(prefix="kpeggold")
    f = os.path.join(d, "in.jpg")
    open(f, "wb").write(data)
    out = subprocess.run([T.REF_BIN, "status", f], capture_output=True, text=True, timeout=600)
    try:
        return json.loads(out.stdout.strip().splitlines()[-1])["status"]
    except Exception:
        return "CRASH(rc=%d)" % out.returncode


def pil_jpeg(rgb, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", **kw)
    return b.getvalue()


def main():
    assert T.have_ref(), "oracle/_ref/kpeg_ref is missing: run `make -C oracle ref` in the build container"
    manifest = {"decode": {}, "status": {}}
    rng = np.random.default_rng(20261004)

    def add_decode(name, data, keep_ppm):
        info, rgb = T.ref_decode(data)
        assert info["status"] == "DECODE_DONE", (name, info)
        open(os.path.join(HERE, name + ".jpg"), "wb").write(data)
        ppm = T.ppm_bytes(rgb)
        assert sha(ppm) == info["ppm_sha256"]
        manifest["decode"][name] = {"width": info["width"], "height": info["height"], "jpg_sha256": sha(data),
                                    "ppm_sha256": info["ppm_sha256"], "ppm_file": keep_ppm}
        if keep_ppm:
            open(os.path.join(HERE, name + ".ppm"), "wb").write(ppm)

    # synthetic (tools/kpeg_synth.c)
    add_decode("synth_8x8_q75", T.synth_jpeg(8, 8, seed=1), True)
    add_decode("synth_16x8_q90", T.synth_jpeg(16, 8, seed=2, quality=90), True)
    add_decode("synth_64x64_q75", T.synth_jpeg(64, 64, seed=3), True)
    add_decode("synth_136x40_q50", T.synth_jpeg(136, 40, seed=4, quality=50, sigma=12.0), True)
    add_decode("synth_96x64_q95_noise", T.synth_jpeg(96, 64, seed=5, quality=95, sigma=0.0, mode=1), False)
    add_decode("synth_256x128_q30", T.synth_jpeg(256, 128, seed=6, quality=30, sigma=3.0), False)
    # Pillow encodings: other quantiser tables, optimised Huffman tables, flat and saturated content
    yy, xx = np.mgrid[0:64, 0:96]
    grad = np.stack([(xx * 255 // 95), (yy * 255 // 63), ((xx + yy) * 255 // 158)], -1).astype(np.uint8)
    add_decode("pil_96x64_q85", pil_jpeg(grad, quality=85, subsampling=0), True)
    add_decode("pil_96x64_q60_opt", pil_jpeg(grad, quality=60, subsampling=0, optimize=True), True)
    noise = rng.integers(0, 256, size=(48, 72, 3), dtype=np.uint8)
    add_decode("pil_72x48_q92_noise_opt", pil_jpeg(noise, quality=92, subsampling=0, optimize=True), False)
    sat = np.zeros((32, 32, 3), np.uint8)
    sat[:, :16] = (255, 0, 0)
    sat[:16, 16:] = (0, 0, 255)
    sat[16:, 16:] = (255, 255, 255)
    add_decode("pil_32x32_saturated", pil_jpeg(sat, quality=75, subsampling=0), True)
    grey = np.full((16, 24, 3), 128, np.uint8)
    add_decode("pil_24x16_grey", pil_jpeg(grey, quality=75, subsampling=0), True)

    # lena: SHA only (input read from /root/reference when present)
    lena = open("/root/reference/misc/images/lena.jpg", "rb").read()
    info, _ = T.ref_decode(lena)
    manifest["lena"] = {"jpg_sha256": sha(lena), "ppm_sha256": info["ppm_sha256"], "width": info["width"], "height": info["height"],
                        "path": "/root/reference/misc/images/lena.jpg"}

    # accept / reject behaviour
    base = T.synth_jpeg(16, 16, seed=9)
    sos = base.find(b"\xff\xda")

    def add_status(name, data):
        open(os.path.join(HERE, name + ".jpg"), "wb").write(data)
        manifest["status"][name] = {"jpg_sha256": sha(data), "status": ref_status(data)}

    add_status("rej_dri", T.synth_jpeg(16, 16, seed=9, restart_interval=2))
    add_status("rej_app1", base[:2] + b"\xff\xe1\x00\x08Exif\x00\x00" + base[2:])
    add_status("rej_trailing_bytes", base + b"\x00\x00")
    add_status("ok_trailing_ff", base + b"\xff")
    add_status("ok_comment", base[:2] + b"\xff\xfe\x00\x07hello" + base[2:])
    add_status("rej_comment_with_ff", base[:2] + b"\xff\xfe\x00\x05a\xffb" + base[2:])
    add_status("rej_progressive", pil_jpeg(grad, quality=75, subsampling=0, progressive=True))
    add_status("rej_420", pil_jpeg(grad, quality=75, subsampling=2))
    add_status("rej_garbage", b"\x00\x01\x02\x03")
    # (an empty file makes the reference allocate an image of uninitialised size: not run)
    add_status("ok_no_eoi", base[:-2])
    add_status("rej_sof1", base.replace(b"\xff\xc0", b"\xff\xc1", 1))
    add_status("rej_ff_fill_before_marker", base[:sos] + b"\xff" + base[sos:])

    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(manifest["decode"]), "decode fixtures,", len(manifest["status"]), "status fixtures")
    for k, v in manifest["status"].items():
        print("  %-28s %s" % (k, v["status"]))


if __name__ == "__main__":
    main()
