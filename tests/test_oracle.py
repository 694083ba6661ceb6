"""The oracle (oracle/kpeg_oracle.c) against the golden fixtures produced by the real reference
(tests/golden/make_golden.py) and, where the build container has it, against the reference binary
itself.  CPU only."""
import json
import os

import numpy as np
import pytest

import kpeg_testlib as T

MAN = json.load(open(os.path.join(T.GOLDEN, "manifest.json")))
STATUS_CODE = {"SUCCESS": 0, "TERMINATE": 1, "ERROR": 2, "DECODE_INCOMPLETE": 3, "DECODE_DONE": 4}


def _read(name):
    return open(os.path.join(T.GOLDEN, name + ".jpg"), "rb").read()


@pytest.mark.parametrize("name", sorted(MAN["decode"]))
def test_oracle_matches_reference_golden(name):
    g = MAN["decode"][name]
    data = _read(name)
    assert T.sha256(data) == g["jpg_sha256"]
    st, rgb = T.oracle_decode(data, nthreads=2)
    assert st == T.DECODE_DONE
    assert rgb.shape == (g["height"], g["width"], 3)
    ppm = T.ppm_bytes(rgb)
    assert T.sha256(ppm) == g["ppm_sha256"]
    if g["ppm_file"]:
        assert ppm == open(os.path.join(T.GOLDEN, name + ".ppm"), "rb").read()


@pytest.mark.parametrize("name", sorted(MAN["status"]))
def test_oracle_parser_status_matches_reference(name):
    g = MAN["status"][name]
    p = T.oracle_parse(_read(name))
    assert p.status == STATUS_CODE[g["status"]], (name, g["status"], p.status)


def test_lena_sha256():
    """The reference's own sample image."""
    g = MAN["lena"]
    assert g["ppm_sha256"] == "064dace1c86b7d2d887ae5d2b76fc53444136e155cc45cf2bad867fcdb13078d"
    data = open(os.path.join(T.GOLDEN, g["fixture"]), "rb").read()   # the reference's sample, committed as a data fixture
    assert T.sha256(data) == g["jpg_sha256"]
    st, rgb = T.oracle_decode(data, nthreads=2)
    assert st == T.DECODE_DONE
    assert T.sha256(T.ppm_bytes(rgb)) == g["ppm_sha256"]


@pytest.mark.skipif(not T.have_ref(), reason="real reference binary only exists in the build container")
@pytest.mark.parametrize("w,h,q,sigma,mode,seed", [(40, 24, 75, 6.0, 0, 21), (72, 16, 35, 20.0, 0, 22), (32, 32, 98, 0.0, 1, 23),
                                                   (320, 64, 75, 6.0, 0, 24)])
def test_oracle_matches_live_reference(w, h, q, sigma, mode, seed):
    data = T.synth_jpeg(w, h, seed=seed, quality=q, sigma=sigma, mode=mode)
    info, want = T.ref_decode(data)
    assert info["status"] == "DECODE_DONE"
    st, got = T.oracle_decode(data, nthreads=2)
    assert st == T.DECODE_DONE
    assert np.array_equal(got, want)


def test_unstuff_fast_form_equals_literal_form():
    rng = np.random.default_rng(1)
    L = T.oracle()
    for trial in range(200):
        n = int(rng.integers(1, 40))
        a = rng.choice(np.array([0x00, 0xFF, 0x12, 0xD9], np.uint8), size=n).astype(np.uint8)
        o1 = np.empty(n, np.uint8)
        o2 = np.empty(n, np.uint8)
        n1 = L.kpeg_oracle_unstuff(a.ctypes.data, n, o1.ctypes.data)
        n2 = L.kpeg_oracle_unstuff_literal(a.ctypes.data, n, o2.ctypes.data)
        assert n1 == n2 and np.array_equal(o1[:n1], o2[:n2]), a.tobytes().hex()


def test_unstuff_tail_rule():
    # byteStuffScanData never erases the very last byte (Decoder.cpp:637)
    assert T.oracle_unstuff(b"\x12\xff\x00") == b"\x12\xff\x00"
    assert T.oracle_unstuff(b"\x12\xff\x00\x34") == b"\x12\xff\x34"
    assert T.oracle_unstuff(b"\xff\x00\x00\x01") == b"\xff\x00\x01"
    assert T.oracle_unstuff(b"\xff\xff\x00\x01") == b"\xff\xff\x01"


def test_quirk_q1_dc_eob_drops_ac():
    """A block whose DC difference is coded as symbol 0x00 loses its AC terms."""
    w, h = 16, 8
    coef = np.zeros((2, 3, 64), np.int16)
    coef[0, 0, 0] = 10
    coef[0, 0, 1] = 5
    coef[1, 0, 0] = 10   # same DC as the previous Y block: difference 0 -> "EOB"
    coef[1, 0, 1] = 7    # ... so this AC term must vanish
    coef[1, 1, 0] = 3
    coef[1, 1, 2] = -4   # Cb DC differs from the previous Cb block (0 -> 3): kept
    q = np.full(64, 8, np.uint16)
    data = T.encode_coefs(coef, w, h, q, q)
    p = T.oracle_parse(data)
    rc, got = T.oracle_entropy(p)
    assert rc == 0
    assert got[0, 0, 1] == 5 and got[1, 0, 0] == 10 and got[1, 0, 1] == 0 and got[1, 1, 2] == -4
    if T.have_ref():
        info, want = T.ref_decode(data)
        assert np.array_equal(T.oracle_idct_colour(got, p.qt, w, h), want)


def test_colour_known_answer():
    """colorTest() of the reference's main.cpp:328-346: Y=383, Cb=Cr=128 -> (255,255,255)."""
    # one 8x8 MCU whose Y block is DC-only with F/8 + 128 = 383: F = 2040 = 255 * 8
    coef = np.zeros((1, 3, 64), np.int16)
    coef[0, 0, 0] = 255
    qt = np.full((2, 64), 8, np.uint16)
    rgb = T.oracle_idct_colour(coef, qt, 8, 8, 1)
    assert (rgb == 255).all()


def test_ppm_header():
    hdr = T.ppm_bytes(np.zeros((8, 16, 3), np.uint8))[:-8 * 16 * 3]
    assert hdr == b"P6\n# PPM dump created using libKPEG: https://github.com/TheIllusionistMirage/libKPEG\n16 8\n255\n"


# ---- BASELINE-sized and natural-content pins (tests/golden/make_golden_large.py: hashes from the real reference) ----
LARGE = json.load(open(os.path.join(T.GOLDEN, "manifest_large.json")))


@pytest.mark.parametrize("name", sorted(LARGE["natural"]))
def test_oracle_matches_reference_on_photographs(name):
    """Pillow encodings of photographs, 1-5 bits per pixel: other quantisers, optimised Huffman tables,
    natural coefficient statistics (multi-round re-synchronisation on the GPU side)."""
    g = LARGE["natural"][name]
    data = _read(name)
    assert T.sha256(data) == g["jpg_sha256"]
    st, rgb = T.oracle_decode(data, nthreads=4)
    assert st == T.DECODE_DONE and rgb.shape == (g["height"], g["width"], 3)
    assert T.sha256(T.ppm_bytes(rgb)) == g["ppm_sha256"]


def test_oracle_matches_reference_at_1080p():
    """BASELINE config 2's input: the generator reproduces the pinned file byte for byte and the oracle the
    reference's PPM (the 8K and 16384x16384 pins are checked on the GPU side, tests/test_gpu_large.py; the
    generator script itself asserted oracle == reference on both)."""
    g = LARGE["synth"]["1920x1080_seed1234"]
    data = T.synth_jpeg(g["width"], g["height"], seed=g["seed"])
    assert T.sha256(data) == g["jpg_sha256"]
    st, rgb = T.oracle_decode(data, nthreads=8)
    assert st == T.DECODE_DONE
    assert T.sha256(T.ppm_bytes(rgb)) == g["ppm_sha256"]
    assert T.sha256(rgb.tobytes()) == g["rgb_sha256"]
