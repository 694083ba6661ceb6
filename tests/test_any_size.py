"""The any-size extension (widths / heights that are not multiples of 8), CPU side.

PARITY UNPINNED as far as the reference's own behaviour on such files goes: it decodes (w * h) / 64 MCUs and tiles
ceil(w / 8) * ceil(h / 8) of them, reading past its MCU vector (undefined behaviour; it happens to answer DECODE_DONE
with garbage in the missing MCUs).  What CAN be pinned is everything but the crop: a committed fixture whose SOF0
header is patched to a smaller size holds the same entropy-coded data, so the any-size decode must equal the top-left
crop of the reference's own output for the unpatched file (tests/golden/*.ppm, hashes in manifest.json) -- the crop
being Image::createImageFromMCUs' (Image.cpp:73-84)."""
import io
import os

import numpy as np
import pytest

import kpeg_testlib as T

GOLD = T.GOLDEN
CASES = [("pil_96x64_q85", 93, 59), ("pil_96x64_q85", 89, 64), ("pil_96x64_q85", 96, 57), ("pil_96x64_q60_opt", 95, 63),
         ("synth_136x40_q50", 129, 33), ("synth_16x8_q90", 9, 1), ("synth_8x8_q75", 1, 1), ("pil_32x32_saturated", 31, 25)]


def patched(name, w, h):
    """The fixture with its SOF0 size replaced: (file bytes, the reference's pixels for the original, cropped)."""
    d = bytearray(open(os.path.join(GOLD, name + ".jpg"), "rb").read())
    i = d.find(b"\xff\xc0")
    H0, W0 = (d[i + 5] << 8) | d[i + 6], (d[i + 7] << 8) | d[i + 8]
    assert (W0 + 7) // 8 == (w + 7) // 8 and (H0 + 7) // 8 == (h + 7) // 8, "same MCU grid"
    d[i + 5:i + 9] = bytes([h >> 8, h & 255, w >> 8, w & 255])
    ppm = open(os.path.join(GOLD, name + ".ppm"), "rb").read()
    ref = np.frombuffer(ppm[len(T.ppm_header(W0, H0)):], np.uint8).reshape(H0, W0, 3)
    return bytes(d), ref[:h, :w].copy()


@pytest.mark.parametrize("name,w,h", CASES)
def test_oracle_equals_the_cropped_reference_output(name, w, h):
    data, want = patched(name, w, h)
    st, got = T.oracle_decode_any_size(data)
    assert st == T.DECODE_DONE and got.shape == (h, w, 3)
    assert np.array_equal(got, want)
    # without the extension the restatement answers as before
    assert T.oracle_decode(data)[0] != T.DECODE_DONE


def test_parser_needs_the_flag():
    import libkpeg_amd as K
    data, _ = patched("pil_96x64_q85", 93, 59)
    rc, frame, scan = K.host_parse(data)
    assert rc == K.ERROR and frame is None          # the product's answer without the extension (decodeScanData)
    rc, frame, scan = K.host_parse(data, allow_any_size=True)
    assert rc == K.DECODE_DONE and (frame.width, frame.height) == (93, 59)
    whole = open(os.path.join(GOLD, "pil_96x64_q85.jpg"), "rb").read()
    assert np.array_equal(scan, K.host_parse(whole)[2])


def test_pillow_files_of_odd_sizes():
    """Real encodings (the encoder pads by edge replication): the restatement against Pillow's decoder, compared where the
    reference's quirk Q1 does not bite (blocks whose differences stay within the two IDCTs' tolerance must be the
    overwhelming majority on a smooth picture)."""
    Image = pytest.importorskip("PIL.Image")
    ph = np.asarray(Image.open(os.path.join(GOLD, "nat_flower_640x424_q75_opt.jpg")).convert("RGB"))
    for (w, h) in [(637, 421), (333, 200), (17, 23)]:
        buf = io.BytesIO()
        Image.fromarray(ph[:h, :w]).save(buf, "JPEG", quality=92, subsampling=0)
        st, got = T.oracle_decode_any_size(buf.getvalue())
        assert st == T.DECODE_DONE and got.shape == (h, w, 3)
        pil = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")).astype(int)
        d = np.abs(got.astype(int) - pil).max(axis=2)
        assert (d <= 3).mean() > 0.9, (w, h, (d <= 3).mean())
        # the last column and row are real picture content, not padding
        assert (d[:, -1] <= 3).mean() > 0.8 and (d[-1, :] <= 3).mean() > 0.8
