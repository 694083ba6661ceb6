"""Host-side C++ API (kpeg::JPEGDecoder marker parser, HuffmanTree, helpers) through the C shim.
CPU only: nothing here touches the GPU."""
import ctypes
import json
import os

import numpy as np
import pytest

import kpeg_testlib as T

MAN = json.load(open(os.path.join(T.GOLDEN, "manifest.json")))
STATUS_CODE = {"SUCCESS": 0, "TERMINATE": 1, "ERROR": 2, "DECODE_INCOMPLETE": 3, "DECODE_DONE": 4}


@pytest.fixture(scope="module")
def K():
    import libkpeg_amd
    libkpeg_amd.load_host()
    return libkpeg_amd


def _read(name):
    return open(os.path.join(T.GOLDEN, name + ".jpg"), "rb").read()


@pytest.mark.parametrize("name", sorted(MAN["status"]))
def test_parser_status_matches_reference(K, name):
    rc, frame, scan = K.host_parse(_read(name))
    assert rc == STATUS_CODE[MAN["status"][name]["status"]], (name, MAN["status"][name]["status"], rc)


@pytest.mark.parametrize("name", sorted(MAN["decode"]))
def test_parser_tables_match_oracle(K, name):
    data = _read(name)
    rc, frame, scan = K.host_parse(data)
    assert rc == K.DECODE_DONE
    p = T.oracle_parse(data)
    assert (frame.width, frame.height) == (p.width, p.height)
    assert scan.tobytes() == p.scan
    for t in range(2):
        assert [frame.qt[t][k] for k in range(64)] == [int(v) for v in p.qt[t]]
    for c in range(2):
        for i in range(2):
            counts, symbols, defined = p.dht[c][i]
            assert defined
            n = sum(counts)
            assert bytes(frame.dht[c][i].counts) == counts
            assert bytes(frame.dht[c][i].symbols)[:n] == symbols[:n]
    assert frame.restart_interval == 0


def test_dri_is_rejected_unless_the_extension_is_on(K):
    data = T.synth_jpeg(64, 32, seed=1, restart_interval=8)
    assert K.host_parse(data)[0] == K.ERROR
    rc, frame, scan = K.host_parse(data, allow_dri=True)
    assert rc == K.DECODE_DONE and frame.restart_interval == 8
    offs = K.restart_offsets(scan)
    assert len(offs) == (64 // 8) * (32 // 8) // 8 - 1
    for i, o in enumerate(offs):
        assert scan[int(o)] == 0xFF and scan[int(o) + 1] == 0xD0 + (i & 7)


def test_stripe_ranges_cover_the_scan(K):
    w, h = 64, 64
    data = T.synth_jpeg(w, h, seed=2, restart_interval=w // 8)
    rc, frame, scan = K.host_parse(data, allow_dri=True)
    ranges = K.stripe_ranges(scan, h // 8, w // 8, w // 8, 4)
    assert [r[:2] for r in ranges] == [(0, 2), (2, 2), (4, 2), (6, 2)]
    assert ranges[0][2] == 0 and ranges[-1][3] == len(scan)
    for a, b in zip(ranges, ranges[1:]):
        assert b[2] == a[3] + 2  # exactly one RSTn between neighbouring stripes
        assert scan[a[3]] == 0xFF and 0xD0 <= scan[a[3] + 1] <= 0xD7


# ---- known answers captured from the reference's dormant test functions ----
KAT_COUNTS = [0, 2, 1, 3, 3, 1, 0, 0, 0, 3, 2, 0, 1, 0, 2, 1]
KAT_SYMBOLS = [0x01, 0x02, 0x03, 0x11, 0x04, 0x00, 0x05, 0x21, 0x12, 0x07, 0xA0, 0xA1, 0xA3, 0xC3, 0x14, 0x27, 0x3A, 0x4A, 0x56]
KAT_CODES = {0x01: "00", 0x02: "01", 0x03: "100", 0x11: "1010", 0x04: "1011", 0x00: "1100", 0x05: "11010", 0x21: "11011",
             0x12: "11100", 0x07: "111010", 0xA0: "1110110000", 0xA1: "1110110001", 0xA3: "1110110010", 0xC3: "11101100110",
             0x14: "11101100111", 0x27: "1110110100000", 0x3A: "111011010000100", 0x4A: "111011010000101",
             0x56: "1110110100001100"}


def _contains(K, bits, counts=KAT_COUNTS, symbols=KAT_SYMBOLS):
    H = K.load_host()
    H.kpeg_host_huffman_contains.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    out = ctypes.create_string_buffer(16)
    n = H.kpeg_host_huffman_contains(bytes(counts), bytes(symbols), bits.encode(), out, 16)
    assert n >= 0
    return out.value.decode()


def test_huffman_tree_known_answers(K):
    # huffmanTreeTest(), main.cpp:142-203
    assert _contains(K, "100") == "3"
    assert _contains(K, "1100") == "EOB"
    assert _contains(K, "101") == ""
    assert _contains(K, "1111111111111111") == ""
    assert _contains(K, "111010") == "7"
    assert _contains(K, "111011010000101") == "74"
    assert _contains(K, "1110110100001100") == "86"


def test_huffman_tree_assigns_canonical_codes(K):
    for sym, code in KAT_CODES.items():
        assert _contains(K, code) == ("EOB" if sym == 0 else str(sym)), (hex(sym), code)
        if len(code) > 1:
            assert _contains(K, code[:-1]) == ""  # prefix-free


def test_bitstring_helpers(K):
    H = K.load_host()
    H.kpeg_host_bitstring_to_value.argtypes = [ctypes.c_char_p]
    f = lambda s: H.kpeg_host_bitstring_to_value(s.encode())
    assert f("") == 0 and f("1") == 1 and f("0") == -1 and f("10") == 2 and f("01") == -2 and f("00") == -3
    assert f("11111111111") == 2047 and f("00000000000") == -2047
    H.kpeg_host_value_to_bitstring.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    out = ctypes.create_string_buffer(32)
    for v in [1, -1, 2, -2, 5, -5, 255, -255, 1023, -1024 + 1]:
        H.kpeg_host_value_to_bitstring(v, out, 32)
        assert f(out.value.decode()) == v


def test_is_valid_filename_quirks(K):
    H = K.load_host()
    H.kpeg_host_is_valid_filename.argtypes = [ctypes.c_char_p]
    ok = lambda s: bool(H.kpeg_host_is_valid_filename(s.encode()))
    assert ok("a.jpg") and ok("dir/x.y.jpg")
    assert not ok("a.jpeg")          # rejected by the reference's length test (Utility.hpp:23-37)
    assert not ok("a.jpg.bak") and not ok("a.png") and not ok("jpg")


def test_batch_front_end_without_a_gpu_fails_loudly(K, tmp_path):
    """`kpeg --batch` parses on the host and decodes on the GPU only: without a gfx950 device every accepted file is
    reported as failed (there is no CPU decode path), rejected files as rejected, nothing is written, the exit code is non-zero."""
    import shutil, subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_decode.py covers the batch front end")
    d = tmp_path / "in"
    d.mkdir()
    shutil.copy(os.path.join(T.GOLDEN, "synth_64x64_q75.jpg"), d / "a.jpg")
    shutil.copy(os.path.join(T.GOLDEN, "rej_dri.jpg"), d / "rej_dri.jpg")
    (d / "note.txt").write_text("not an image")
    out = subprocess.run([K.CLI, "--batch", str(d)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert "0 PPM written, 1 rejected, 1 failed" in out.stdout, out.stdout[-600:]
    assert out.returncode != 0
    assert not os.path.exists(d / "a.ppm")
