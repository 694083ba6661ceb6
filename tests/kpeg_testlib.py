"""Test infrastructure: ctypes access to the oracle (oracle/), the synthetic JPEG generator
(tools/) and the real reference binary (oracle/_ref, build container only).

Nothing here is product code and nothing in libkpeg_amd imports it.
"""
import ctypes
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libkpeg_oracle.so")
SYNTH_SO = os.path.join(ROOT, "tools", "libkpeg_synth.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "kpeg_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden")

DECODE_DONE = 4
OUT_OF_CONTRACT = 100


def ensure_built():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ROOT, "oracle", "kpeg_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])
    if not os.path.exists(SYNTH_SO) or os.path.getmtime(SYNTH_SO) < os.path.getmtime(os.path.join(ROOT, "tools", "kpeg_synth.c")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "all"])


def have_ref():
    return os.path.exists(REF_BIN)


# ------------------------------------------------------------------ oracle
class OracleDht(ctypes.Structure):
    _fields_ = [("counts", ctypes.c_uint8 * 16), ("symbols", ctypes.c_uint8 * 256),
                ("nsymbols", ctypes.c_int), ("defined", ctypes.c_int)]


class OracleJfif(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("nqt", ctypes.c_int),
                ("qt", (ctypes.c_uint16 * 64) * 4), ("dht", (OracleDht * 2) * 2),
                ("scan", ctypes.POINTER(ctypes.c_uint8)), ("scan_len", ctypes.c_size_t), ("saw_sos", ctypes.c_int)]


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        ensure_built()
        L = ctypes.CDLL(ORACLE_SO)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.kpeg_oracle_parse.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(OracleJfif)]
        L.kpeg_oracle_jfif_free.argtypes = [ctypes.POINTER(OracleJfif)]
        L.kpeg_oracle_unstuff.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.kpeg_oracle_unstuff.restype = ctypes.c_size_t
        L.kpeg_oracle_unstuff_literal.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.kpeg_oracle_unstuff_literal.restype = ctypes.c_size_t
        L.kpeg_oracle_entropy_decode.argtypes = [ctypes.POINTER(OracleJfif), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                                 ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.kpeg_oracle_entropy_decode_rst.argtypes = [ctypes.POINTER(OracleJfif), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                                     ctypes.c_uint32, ctypes.c_void_p]
        L.kpeg_oracle_idct_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.kpeg_oracle_idct_block.restype = None
        L.kpeg_oracle_idct_colour.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
        L.kpeg_oracle_idct_colour.restype = None
        L.kpeg_oracle_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(u8p), ctypes.POINTER(ctypes.c_uint32),
                                         ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        L.kpeg_oracle_ppm_header.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
        L.kpeg_oracle_ppm_header.restype = ctypes.c_size_t
        L.kpeg_oracle_cos_table.argtypes = [ctypes.c_void_p]
        L.kpeg_oracle_cos_table.restype = None
        L.kpeg_oracle_zz_to_rowmajor.argtypes = [ctypes.c_int]
        L.free = ctypes.CDLL(None).free
        L.free.argtypes = [ctypes.c_void_p]
        _oracle = L
    return _oracle


ZZ = None


def zz_table():
    global ZZ
    if ZZ is None:
        ZZ = np.array([oracle().kpeg_oracle_zz_to_rowmajor(k) for k in range(64)], dtype=np.int64)
    return ZZ


class Parsed:
    """Result of the oracle's marker parser."""

    def __init__(self, status, width=0, height=0, qt=None, dht=None, scan=b"", nqt=0):
        self.status, self.width, self.height, self.qt, self.dht, self.scan, self.nqt = status, width, height, qt, dht, scan, nqt


def oracle_parse(data):
    L = oracle()
    j = OracleJfif()
    st = L.kpeg_oracle_parse(data, len(data), ctypes.byref(j))
    scan = bytes(bytearray(j.scan[i] for i in range(0))) if False else (ctypes.string_at(j.scan, j.scan_len) if j.scan else b"")
    qt = np.array([[j.qt[t][k] for k in range(64)] for t in range(4)], dtype=np.uint16)
    dht = [[(bytes(j.dht[c][i].counts), bytes(j.dht[c][i].symbols), j.dht[c][i].defined) for i in range(2)] for c in range(2)]
    p = Parsed(st, j.width, j.height, qt, dht, scan, j.nqt)
    p._raw = j  # keeps tables for entropy decode; scan pointer is freed below, so re-attach a copy
    L.kpeg_oracle_jfif_free(ctypes.byref(j))
    return p


def _jfif_from_parsed(p):
    j = OracleJfif()
    j.width, j.height, j.nqt = p.width, p.height, p.nqt
    for t in range(4):
        for k in range(64):
            j.qt[t][k] = int(p.qt[t][k])
    for c in range(2):
        for i in range(2):
            counts, symbols, defined = p.dht[c][i]
            for k in range(16):
                j.dht[c][i].counts[k] = counts[k]
            for k in range(256):
                j.dht[c][i].symbols[k] = symbols[k]
            j.dht[c][i].defined = defined
            j.dht[c][i].nsymbols = sum(counts)
    return j


def oracle_unstuff(scan):
    L = oracle()
    a = np.frombuffer(scan, np.uint8).copy()
    out = np.empty(max(len(a), 1), np.uint8)
    n = L.kpeg_oracle_unstuff(a.ctypes.data, len(a), out.ctypes.data)
    return out[:n].tobytes()


def oracle_entropy(p, restart_interval=0):
    """Quantised coefficients [nmcu,3,64] in zig-zag order (absolute DC, Q1 applied)."""
    L = oracle()
    j = _jfif_from_parsed(p)
    nmcu = (p.width // 8) * (p.height // 8)
    coef = np.zeros((nmcu, 3, 64), np.int16)
    if restart_interval:
        a = np.frombuffer(p.scan, np.uint8).copy()
        rc = L.kpeg_oracle_entropy_decode_rst(ctypes.byref(j), a.ctypes.data, len(a), nmcu, restart_interval, coef.ctypes.data)
    else:
        bits = np.frombuffer(oracle_unstuff(p.scan), np.uint8).copy()
        used = ctypes.c_uint64()
        rc = L.kpeg_oracle_entropy_decode(ctypes.byref(j), bits.ctypes.data, len(bits), nmcu, coef.ctypes.data, ctypes.byref(used))
    return rc, coef


def zz_to_natural(coef_zz):
    """[...,64] zig-zag -> [...,8,8] natural (row, col)."""
    out = np.zeros_like(coef_zz)
    out[..., zz_table()] = coef_zz
    return out.reshape(coef_zz.shape[:-1] + (8, 8))


def oracle_idct_colour(coef_zz, qt2, width, height, nthreads=8):
    L = oracle()
    coef = np.ascontiguousarray(coef_zz, np.int16)
    q = np.ascontiguousarray(qt2[:2], np.uint16)
    rgb = np.empty((height, width, 3), np.uint8)
    L.kpeg_oracle_idct_colour(coef.ctypes.data, q.ctypes.data, width, height, rgb.ctypes.data, nthreads)
    return rgb


def oracle_decode(data, nthreads=8):
    """Returns (status, rgb or None)."""
    L = oracle()
    rgb = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    st = L.kpeg_oracle_decode(data, len(data), ctypes.byref(rgb), ctypes.byref(w), ctypes.byref(h), nthreads)
    if st != DECODE_DONE:
        return st, None
    a = np.ctypeslib.as_array(rgb, shape=(h.value, w.value, 3)).copy()
    L.free(rgb)
    return st, a


def oracle_decode_any_size(data, nthreads=8):
    """The any-size extension (oracle/kpeg_oracle.c: parity unpinned).  Returns (status, rgb or None)."""
    L = oracle()
    L.kpeg_oracle_decode_any_size.restype = ctypes.c_int
    L.kpeg_oracle_decode_any_size.argtypes = L.kpeg_oracle_decode.argtypes
    rgb = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    st = L.kpeg_oracle_decode_any_size(data, len(data), ctypes.byref(rgb), ctypes.byref(w), ctypes.byref(h), nthreads)
    if st != DECODE_DONE:
        return st, None
    a = np.ctypeslib.as_array(rgb, shape=(h.value, w.value, 3)).copy()
    L.free(rgb)
    return st, a


def oracle_decode_420(data, nthreads=8):
    """The 4:2:0 extension (oracle/kpeg_oracle.c: parity unpinned).  Returns (status, rgb or None)."""
    L = oracle()
    L.kpeg_oracle_decode_420.restype = ctypes.c_int
    L.kpeg_oracle_decode_420.argtypes = L.kpeg_oracle_decode.argtypes
    rgb = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    st = L.kpeg_oracle_decode_420(data, len(data), ctypes.byref(rgb), ctypes.byref(w), ctypes.byref(h), nthreads)
    if st != DECODE_DONE:
        return st, None
    a = np.ctypeslib.as_array(rgb, shape=(h.value, w.value, 3)).copy()
    L.free(rgb)
    return st, a


def oracle_decode_gray(data, nthreads=8):
    """The one-component extension (oracle/kpeg_oracle.c: parity unpinned, the reference cannot decode these).
    Returns (status, rgb or None)."""
    L = oracle()
    L.kpeg_oracle_decode_gray.restype = ctypes.c_int
    L.kpeg_oracle_decode_gray.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                          ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
    rgb = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    st = L.kpeg_oracle_decode_gray(data, len(data), ctypes.byref(rgb), ctypes.byref(w), ctypes.byref(h), nthreads)
    if st != DECODE_DONE:
        return st, None
    a = np.ctypeslib.as_array(rgb, shape=(h.value, w.value, 3)).copy()
    L.free(rgb)
    return st, a


def oracle_decode_rst(data, restart_interval, nthreads=8):
    """Oracle for DRI streams (rejected by the reference): strip the DRI segment for parsing,
    decode every restart interval as its own stream."""
    i = data.find(b"\xff\xdd\x00\x04")
    assert i > 0
    stripped = data[:i] + data[i + 6:]
    p = oracle_parse(stripped)
    assert p.status == DECODE_DONE, p.status
    rc, coef = oracle_entropy(p, restart_interval)
    assert rc == 0, rc
    return oracle_idct_colour(coef, p.qt, p.width, p.height, nthreads), p, coef


def ppm_header(w, h):
    L = oracle()
    buf = ctypes.create_string_buffer(256)
    n = L.kpeg_oracle_ppm_header(w, h, buf, 256)
    return buf.raw[:n]


def ppm_bytes(rgb):
    return ppm_header(rgb.shape[1], rgb.shape[0]) + rgb.tobytes()


def sha256(b):
    return hashlib.sha256(b).hexdigest()


# ------------------------------------------------------------------ synthetic JPEGs
_synth = None


def synth():
    global _synth
    if _synth is None:
        ensure_built()
        S = ctypes.CDLL(SYNTH_SO)
        S.kpeg_synth_jpeg.restype = ctypes.c_size_t
        S.kpeg_synth_jpeg.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32,
                                      ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
        S.kpeg_synth_jpeg_rows.restype = ctypes.c_size_t
        S.kpeg_synth_jpeg_rows.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int,
                                           ctypes.c_uint32, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
        S.kpeg_synth_encode_rgb.restype = ctypes.c_size_t
        S.kpeg_synth_encode_rgb.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32,
                                            ctypes.c_void_p, ctypes.c_size_t]
        S.kpeg_synth_encode_coefs.restype = ctypes.c_size_t
        S.kpeg_synth_encode_coefs.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
        _synth = S
    return _synth


def synth_jpeg(w, h, seed=1234, quality=75, restart_interval=0, sigma=6.0, mode=0, y0=0):
    """SURVEY.md 8(d) synthetic 4:4:4 baseline JPEG (mode 1 = dense uniform noise).
    y0: first row of the (virtual) full image this file covers."""
    cap = w * h * 3 + (w * h) // 2 + 65536
    buf = np.empty(cap, np.uint8)
    n = synth().kpeg_synth_jpeg_rows(w, h, y0, seed, quality, restart_interval, sigma, mode, buf.ctypes.data, cap)
    assert n > 0, "synthetic encoder overflow"
    return buf[:n].tobytes()


def encode_rgb(rgb, quality=75, restart_interval=0):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    cap = w * h * 4 + 65536
    buf = np.empty(cap, np.uint8)
    n = synth().kpeg_synth_encode_rgb(rgb.ctypes.data, w, h, quality, restart_interval, buf.ctypes.data, cap)
    assert n > 0
    return buf[:n].tobytes()


def encode_coefs(coef_zz, w, h, ql, qc, restart_interval=0):
    """coef_zz [nmcu,3,64] int16 zig-zag absolute-DC; ql/qc natural-order uint16[64]."""
    coef = np.ascontiguousarray(coef_zz, np.int16)
    ql = np.ascontiguousarray(ql, np.uint16)
    qc = np.ascontiguousarray(qc, np.uint16)
    cap = coef.size * 4 + 65536
    buf = np.empty(cap, np.uint8)
    n = synth().kpeg_synth_encode_coefs(coef.ctypes.data, w, h, ql.ctypes.data, qc.ctypes.data, restart_interval, buf.ctypes.data, cap)
    assert n > 0
    return buf[:n].tobytes()


# ------------------------------------------------------------------ real reference (build container only)
def ref_decode(data):
    """Run the real reference on `data` in a fresh process.  Returns (status_dict, rgb or None)."""
    assert have_ref()
    d = tempfile.mkdtemp(prefix="kpegref")
    try:
        f = os.path.join(d, "in.jpg")
        with open(f, "wb") as fh:
            fh.write(data)
        out = subprocess.run([REF_BIN, "decode", f], capture_output=True, text=True, timeout=3600)
        info = json.loads(out.stdout.strip().splitlines()[-1]) if out.stdout.strip() else {"status": "CRASH", "rc": out.returncode}
        ppm = os.path.join(d, "in.ppm")
        if info.get("status") != "DECODE_DONE" or not os.path.exists(ppm):
            return info, None
        raw = open(ppm, "rb").read()
        parts = raw.split(b"\n", 4)
        w, h = map(int, parts[2].split())
        info["ppm_sha256"] = sha256(raw)
        return info, np.frombuffer(parts[4], np.uint8).reshape(h, w, 3).copy()
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ------------------------------------------------------------------ frame for the product ABI
def make_frame(p, restart_interval=0):
    """kpeg_frame from an oracle parse (tests only; the product's own parser is kpeg::JPEGDecoder)."""
    import libkpeg_amd
    f = libkpeg_amd.Frame()
    f.width, f.height = p.width, p.height
    for t in range(2):
        for k in range(64):
            f.qt[t][k] = int(p.qt[t][k])
    for c in range(2):
        for i in range(2):
            counts, symbols, _ = p.dht[c][i]
            for k in range(16):
                f.dht[c][i].counts[k] = counts[k]
            for k in range(256):
                f.dht[c][i].symbols[k] = symbols[k]
    f.restart_interval = restart_interval
    return f
