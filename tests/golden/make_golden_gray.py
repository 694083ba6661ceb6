#!/usr/bin/env python3
"""tests/golden/make_golden_gray.py -- fixtures for the one-component (grayscale) extension.

    python tests/golden/make_golden_gray.py

The reference cannot decode one-component files (its SOF0 parser reads three component triples whatever the
header says), so these are NOT pinned to it: parity unpinned.  What is committed:
  gray_*.jpg             Pillow ("L" mode) encodings of a synthetic ramp+noise picture and of a photograph
                         (scikit-learn's bundled sample, as for nat_*.jpg), plain, with optimised tables, and with a restart interval of 3 blocks
  manifest_gray.json     per file: SHA-256 of the file, SHA-256 of the oracle's RGB output
                         (kpeg_oracle_decode_gray: the reference's per-block arithmetic on one component), and the
                         largest difference from Pillow's own decoder when this script ran (two different IDCTs:
                         within 2 levels except in the blocks whose coded DC difference is zero, where the reference's quirk Q1
                         drops the AC terms -- the count of such blocks is recorded)
"""
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests"))
import kpeg_testlib as T  # noqa: E402


def sha(b):
    return hashlib.sha256(b).hexdigest()


def pictures():
    rng = np.random.default_rng(11)
    y, x = np.mgrid[0:48, 0:64]
    yield "gray_ramp_64x48_q75", np.clip(2 * x + 3 * y + rng.normal(0, 6, (48, 64)), 0, 255).astype(np.uint8), dict(quality=75)
    yield "gray_noise_32x16_q95", rng.integers(0, 256, (16, 32), dtype=np.uint8), dict(quality=95)
    yield "gray_noise_64x32_q80_rst3", rng.integers(0, 256, (32, 64), dtype=np.uint8), dict(quality=80, restart_marker_blocks=3)
    from sklearn.datasets import load_sample_image
    ph = np.asarray(Image.fromarray(load_sample_image("flower.jpg")).convert("L"))[:424, :640]
    yield "gray_flower_640x424_q80", ph, dict(quality=80)
    yield "gray_flower_320x208_q60_opt", np.asarray(Image.fromarray(ph).resize((320, 212)))[:208], dict(quality=60, optimize=True)


def main():
    man = {}
    for name, px, kw in pictures():
        buf = io.BytesIO()
        Image.fromarray(px, "L").save(buf, "JPEG", **kw)
        data = buf.getvalue()
        st, rgb = T.oracle_decode_gray(data)
        assert st == T.DECODE_DONE, (name, st)
        assert np.array_equal(rgb[..., 0], rgb[..., 1]) and np.array_equal(rgb[..., 0], rgb[..., 2])
        pil = np.asarray(Image.open(io.BytesIO(data)).convert("L")).astype(int)
        d = int(np.abs(rgb[..., 0].astype(int) - pil).max())
        open(os.path.join(HERE, name + ".jpg"), "wb").write(data)
        blk = np.abs(rgb[..., 0].astype(int) - pil).reshape(px.shape[0] // 8, 8, px.shape[1] // 8, 8).max(axis=(1, 3))
        man[name + ".jpg"] = {"blocks_over_2_from_pillow": int((blk > 2).sum()), "blocks": int(blk.size),"jpg_sha256": sha(data), "width": int(px.shape[1]), "height": int(px.shape[0]),
                              "oracle_rgb_sha256": sha(rgb.tobytes()), "max_diff_from_pillow": d}
        print(name, len(data), "bytes, max |oracle - Pillow| =", d)
    json.dump(man, open(os.path.join(HERE, "manifest_gray.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
