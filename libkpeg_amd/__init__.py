"""libkpeg_amd -- Python plumbing over the MI355X-native libKPEG decode path.

The product is the C-ABI library ``libkpeg_hip.so`` (include/kpeg_hip.h) plus the C++ host
mirror of the reference's ``kpeg::JPEGDecoder`` / ``kpeg::Image`` API (``libkpeg.so`` and the
``kpeg`` CLI).  This package only loads those libraries through ``ctypes`` for tests and
``bench.py``; it contains no decode logic and no CPU fallback: if the HIP library is missing
or no gfx950 device is present, it raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.environ.get("KPEG_HIP_LIB") or os.path.join(_HERE, "libkpeg_hip.so")  # override: kernel experiments only
HOST_LIB = os.path.join(_HERE, "libkpeg.so")
CLI = os.path.join(_HERE, "kpeg")

ABI_VERSION = 2
OK = 0
E_ARG, E_DEVICE, E_TABLES, E_STREAM, E_NOMEM, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6


class KpegError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("kpeg_hip error %d: %s" % (code, msg))
        self.code = code


class Dht(ctypes.Structure):
    _fields_ = [("counts", ctypes.c_uint8 * 16), ("symbols", ctypes.c_uint8 * 256)]


class Frame(ctypes.Structure):
    """kpeg_frame (include/kpeg_hip.h)."""
    _fields_ = [
        ("width", ctypes.c_uint32),
        ("height", ctypes.c_uint32),
        ("qt", (ctypes.c_uint16 * 64) * 2),
        ("dht", (Dht * 2) * 2),
        ("restart_interval", ctypes.c_uint32),
        ("components", ctypes.c_uint32),
    ]


class Timings(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in
                ("unstuff_ms", "huff_sync_ms", "huff_scan_ms", "huff_write_ms", "dc_ms", "idct_ms", "total_ms")] + \
               [("sync_rounds", ctypes.c_uint32), ("exact_pixels", ctypes.c_uint32)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


BAND_SINK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)

EXPORTS = [
    "kpeg_hip_abi_version", "kpeg_hip_build_hash", "kpeg_hip_create", "kpeg_hip_destroy", "kpeg_hip_strerror", "kpeg_hip_last_error",
    "kpeg_hip_set_stream", "kpeg_hip_sync", "kpeg_hip_set_profiling", "kpeg_hip_get_timings",
    "kpeg_hip_idct_colour", "kpeg_hip_decode_scan", "kpeg_hip_decode_batch", "kpeg_hip_decode_batch_dev",
    "kpeg_hip_idct_colour_dev", "kpeg_hip_decode_scan_dev", "kpeg_hip_decode_stripe_dev",
    "kpeg_hip_entropy_decode_dev", "kpeg_hip_set_idct_mode", "kpeg_hip_decode_sharded", "kpeg_hip_decode_sharded_dev",
    "kpeg_hip_decode_scan_resident", "kpeg_hip_download_bands", "kpeg_hip_resident_generation",
]

_lib = None


def load_hip():
    """dlopen libkpeg_hip.so and declare prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HIP_LIB):
        raise RuntimeError("%s is missing: run `python -m libkpeg_amd.build` (no CPU fallback exists)" % HIP_LIB)
    # PyTorch ships its own libamdhip64.so.7.  Two HIP runtimes in one process cannot both open
    # the GPU, so when PyTorch is installed it is imported first: libkpeg_hip.so (NEEDED
    # libamdhip64.so.7) then binds to the copy PyTorch has already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = _declare(ctypes.CDLL(HIP_LIB))
    return _lib


def load_variant(path):
    """Another build of the library beside the default one (kernel A/B experiments, tools/k4_ab.py)."""
    load_hip()
    return _declare(ctypes.CDLL(path))


def _declare(L):
    vp, c_int, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    FP = ctypes.POINTER(Frame)
    L.kpeg_hip_abi_version.restype = c_int
    if hasattr(L, "kpeg_hip_build_hash"):
        L.kpeg_hip_build_hash.restype = ctypes.c_char_p
    L.kpeg_hip_create.argtypes = [ctypes.POINTER(vp), c_int]
    L.kpeg_hip_destroy.argtypes = [vp]
    L.kpeg_hip_destroy.restype = None
    L.kpeg_hip_strerror.argtypes = [c_int]
    L.kpeg_hip_strerror.restype = ctypes.c_char_p
    L.kpeg_hip_last_error.argtypes = [vp]
    L.kpeg_hip_last_error.restype = ctypes.c_char_p
    L.kpeg_hip_set_stream.argtypes = [vp, vp]
    L.kpeg_hip_sync.argtypes = [vp]
    L.kpeg_hip_set_profiling.argtypes = [vp, c_int]
    L.kpeg_hip_get_timings.argtypes = [vp, ctypes.POINTER(Timings)]
    L.kpeg_hip_set_idct_mode.argtypes = [vp, c_int]
    L.kpeg_hip_idct_colour.argtypes = [vp, FP, vp, vp]
    L.kpeg_hip_decode_scan.argtypes = [vp, FP, vp, sz, vp]
    L.kpeg_hip_decode_batch.argtypes = [vp, c_int, FP, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp)]
    L.kpeg_hip_decode_batch_dev.argtypes = [vp, c_int, FP, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp)]
    L.kpeg_hip_idct_colour_dev.argtypes = [vp, FP, vp, vp]
    L.kpeg_hip_decode_scan_dev.argtypes = [vp, FP, vp, sz, vp]
    L.kpeg_hip_decode_stripe_dev.argtypes = [vp, FP, vp, sz, ctypes.c_uint32, ctypes.c_uint32, vp]
    L.kpeg_hip_entropy_decode_dev.argtypes = [vp, FP, vp, sz, vp]
    if hasattr(L, "kpeg_hip_download_bands"):
        L.kpeg_hip_decode_scan_resident.argtypes = [vp, FP, vp, sz]
        L.kpeg_hip_download_bands.argtypes = [vp, FP, ctypes.c_uint32, BAND_SINK, vp]
    if hasattr(L, "kpeg_hip_decode_sharded"):   # (reference builds of earlier trees kept for A/B runs lack the newer entries)
        L.kpeg_hip_decode_sharded.argtypes = [ctypes.POINTER(vp), c_int, FP, vp, sz, vp]
        L.kpeg_hip_decode_sharded_dev.argtypes = [ctypes.POINTER(vp), c_int, FP, vp, sz, vp]
    return L


class Context:
    """Thin RAII wrapper over kpeg_hip_ctx."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or load_hip()
        self._h = ctypes.c_void_p()
        rc = self.lib.kpeg_hip_create(ctypes.byref(self._h), device)
        if rc != OK:
            raise KpegError(rc, "kpeg_hip_create(device=%d) failed: %s" % (device, self.lib.kpeg_hip_strerror(rc).decode()))

    def close(self):
        if self._h:
            self.lib.kpeg_hip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise KpegError(rc, "%s (%s)" % (self.lib.kpeg_hip_strerror(rc).decode(),
                                             self.lib.kpeg_hip_last_error(self._h).decode()))

    # -- knobs
    def set_stream(self, hip_stream):
        """hip_stream: a hipStream_t handle of a CREATED stream, e.g. torch.cuda.Stream().cuda_stream.  PyTorch's default
        stream reports handle 0, which the C ABI reads as "the context's own stream" -- NOT ordered with default-stream
        work (a tensor written by a torch op just before the call may not be complete when the kernels read it): refused
        here; run the torch side under `torch.cuda.stream(s)` / `torch.cuda.set_stream(s)` and pass that stream."""
        if not hip_stream:
            raise ValueError("the default stream (handle 0) cannot be selected: use a created stream, or use_own_stream()")
        self._chk(self.lib.kpeg_hip_set_stream(self._h, ctypes.c_void_p(hip_stream)))

    def use_own_stream(self):
        self._chk(self.lib.kpeg_hip_set_stream(self._h, ctypes.c_void_p(0)))

    def set_profiling(self, on):
        self._chk(self.lib.kpeg_hip_set_profiling(self._h, int(bool(on))))

    def set_idct_mode(self, mode):
        self._chk(self.lib.kpeg_hip_set_idct_mode(self._h, mode))

    def sync(self):
        self._chk(self.lib.kpeg_hip_sync(self._h))

    def timings(self):
        t = Timings()
        self._chk(self.lib.kpeg_hip_get_timings(self._h, ctypes.byref(t)))
        return t.asdict()

    # -- host-buffer entry points
    def idct_colour(self, frame, coef):
        """coef: int16 array [nmcu, 3, 8, 8] (natural order, quantised). Returns HxWx3 uint8."""
        coef = np.ascontiguousarray(coef, dtype=np.int16)
        nmcu = (frame.width // 8) * (frame.height // 8)
        assert coef.size == nmcu * 192, (coef.shape, nmcu)
        rgb = np.empty((frame.height, frame.width, 3), np.uint8)
        self._chk(self.lib.kpeg_hip_idct_colour(self._h, ctypes.byref(frame), coef.ctypes.data, rgb.ctypes.data))
        return rgb

    def decode_scan(self, frame, scan):
        scan = np.frombuffer(scan, dtype=np.uint8) if not isinstance(scan, np.ndarray) else scan
        rgb = np.empty((frame.height, frame.width, 3), np.uint8)
        self._chk(self.lib.kpeg_hip_decode_scan(self._h, ctypes.byref(frame), scan.ctypes.data, scan.size, rgb.ctypes.data))
        return rgb

    def decode_scan_banded(self, frame, scan, band_rows=0):
        """kpeg_hip_decode_scan_resident + kpeg_hip_download_bands; returns (HxWx3 array assembled from the bands, band list)."""
        scan = np.frombuffer(scan, dtype=np.uint8) if not isinstance(scan, np.ndarray) else scan
        self._chk(self.lib.kpeg_hip_decode_scan_resident(self._h, ctypes.byref(frame), scan.ctypes.data, scan.size))
        rgb = np.empty((frame.height, frame.width, 3), np.uint8)
        bands = []

        def sink(user, row0, rows, ptr, nbytes):
            rgb[row0:row0 + rows] = np.ctypeslib.as_array(ptr, shape=(nbytes,)).reshape(rows, frame.width, 3)
            bands.append((row0, rows))
            return 0

        cb = BAND_SINK(sink)
        self._chk(self.lib.kpeg_hip_download_bands(self._h, ctypes.byref(frame), band_rows, cb, None))
        return rgb, bands

    def decode_batch(self, frame, scans):
        """scans: list of byte strings / uint8 arrays of one geometry and one set of tables. Returns a list of HxWx3 uint8."""
        n = len(scans)
        scans = [np.frombuffer(s, dtype=np.uint8) if not isinstance(s, np.ndarray) else np.ascontiguousarray(s) for s in scans]
        outs = [np.empty((frame.height, frame.width, 3), np.uint8) for _ in range(n)]
        sp = (ctypes.c_void_p * n)(*[s.ctypes.data for s in scans])
        sl = (ctypes.c_size_t * n)(*[s.size for s in scans])
        op = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        self._chk(self.lib.kpeg_hip_decode_batch(self._h, n, ctypes.byref(frame), sp, sl, op))
        return outs

    def decode_batch_dev(self, frame, d_scans, scan_lens, d_rgbs):
        """device pointers (ints); asynchronous, errors surface at sync()."""
        n = len(d_scans)
        sp = (ctypes.c_void_p * n)(*d_scans)
        sl = (ctypes.c_size_t * n)(*scan_lens)
        op = (ctypes.c_void_p * n)(*d_rgbs)
        self._chk(self.lib.kpeg_hip_decode_batch_dev(self._h, n, ctypes.byref(frame), sp, sl, op))

    # -- device-resident entry points (raw device pointers, e.g. torch tensors' data_ptr())
    def idct_colour_dev(self, frame, d_coef, d_rgb):
        self._chk(self.lib.kpeg_hip_idct_colour_dev(self._h, ctypes.byref(frame), ctypes.c_void_p(d_coef), ctypes.c_void_p(d_rgb)))

    def decode_scan_dev(self, frame, d_scan, scan_len, d_rgb):
        self._chk(self.lib.kpeg_hip_decode_scan_dev(self._h, ctypes.byref(frame), ctypes.c_void_p(d_scan), scan_len,
                                                    ctypes.c_void_p(d_rgb)))

    def decode_stripe_dev(self, frame, d_scan, scan_len, first_mcu_row, mcu_rows, d_rgb):
        self._chk(self.lib.kpeg_hip_decode_stripe_dev(self._h, ctypes.byref(frame), ctypes.c_void_p(d_scan), scan_len,
                                                      first_mcu_row, mcu_rows, ctypes.c_void_p(d_rgb)))

    def entropy_decode_dev(self, frame, d_scan, scan_len, d_coef):
        self._chk(self.lib.kpeg_hip_entropy_decode_dev(self._h, ctypes.byref(frame), ctypes.c_void_p(d_scan), scan_len,
                                                       ctypes.c_void_p(d_coef)))


def decode_sharded(ctxs, frame, scan, d_rgb_root=None):
    """One image over len(ctxs) GPUs from this process (kpeg_hip_decode_sharded / _dev).  scan: host bytes of the whole
    DRI scan.  d_rgb_root: device pointer on ctxs[0]'s GPU, or None for a host result (returned as an array)."""
    lib = ctxs[0].lib
    scan = np.ascontiguousarray(np.frombuffer(scan, np.uint8) if not isinstance(scan, np.ndarray) else scan)
    hs = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    if d_rgb_root is None:
        rgb = np.empty((frame.height, frame.width, 3), np.uint8)
        ctxs[0]._chk(lib.kpeg_hip_decode_sharded(hs, len(ctxs), ctypes.byref(frame), scan.ctypes.data, scan.size, rgb.ctypes.data))
        return rgb
    ctxs[0]._chk(lib.kpeg_hip_decode_sharded_dev(hs, len(ctxs), ctypes.byref(frame), scan.ctypes.data, scan.size, ctypes.c_void_p(d_rgb_root)))
    return None


# ---------------------------------------------------------------------------------------------
# C++ host library (kpeg::JPEGDecoder marker parser) through its C shim, include/kpeg_host.h
PARSE_ALLOW_DRI = 1
PARSE_ALLOW_GRAY = 2
PARSE_ALLOW_ANY_SIZE = 4
PARSE_ALLOW_420 = 8
FRAME_420 = 0x203
SUCCESS, TERMINATE, ERROR, DECODE_INCOMPLETE, DECODE_DONE = 0, 1, 2, 3, 4
_host = None


def load_host():
    global _host
    if _host is not None:
        return _host
    load_hip()  # libkpeg.so links against libkpeg_hip.so; keeps the HIP runtime load order
    if not os.path.exists(HOST_LIB):
        raise RuntimeError("%s is missing: run `python -m libkpeg_amd.build`" % HOST_LIB)
    H = ctypes.CDLL(HOST_LIB)
    H.kpeg_host_parse.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.POINTER(Frame), ctypes.c_void_p,
                                  ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    H.kpeg_host_decode_file.argtypes = [ctypes.c_char_p, ctypes.c_uint]
    H.kpeg_host_restart_offsets.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    H.kpeg_host_restart_offsets.restype = ctypes.c_size_t
    _host = H
    return H


def host_parse(data, allow_dri=False, allow_gray=False, allow_any_size=False, allow_420=False):
    """Run the product's marker parser on an in-memory JFIF file.
    Returns (result_code, Frame or None, scan bytes (numpy uint8) or None)."""
    H = load_host()
    buf = np.frombuffer(data, np.uint8)
    frame = Frame()
    scan = np.empty(buf.size + 1, np.uint8)
    n = ctypes.c_size_t(0)
    rc = H.kpeg_host_parse(buf.ctypes.data, buf.size, (PARSE_ALLOW_DRI if allow_dri else 0) | (PARSE_ALLOW_GRAY if allow_gray else 0) | (PARSE_ALLOW_ANY_SIZE if allow_any_size else 0) | (PARSE_ALLOW_420 if allow_420 else 0),
                           ctypes.byref(frame),
                           scan.ctypes.data, scan.size, ctypes.byref(n))
    if rc != DECODE_DONE:
        return rc, None, None
    return rc, frame, scan[:n.value]


def restart_offsets(scan):
    """Byte offsets of the RSTn markers in a still-stuffed scan."""
    H = load_host()
    scan = np.ascontiguousarray(scan, np.uint8)
    n = H.kpeg_host_restart_offsets(scan.ctypes.data, scan.size, None, 0)
    out = np.empty(n, np.uint64)
    H.kpeg_host_restart_offsets(scan.ctypes.data, scan.size, out.ctypes.data, n)
    return out


def stripe_ranges(scan, total_mcu_rows, mcus_per_row, restart_interval, nstripes):
    """Split a DRI scan into `nstripes` row stripes on restart-interval boundaries.
    Returns [(first_mcu_row, mcu_rows, byte_begin, byte_end)] -- byte range of each stripe's
    intervals inside `scan` (RSTn between a stripe's own intervals included, the one that
    separates it from the next stripe excluded)."""
    assert restart_interval > 0 and (restart_interval % mcus_per_row == 0 or mcus_per_row % restart_interval == 0)
    offs = restart_offsets(scan)
    nint = (total_mcu_rows * mcus_per_row + restart_interval - 1) // restart_interval
    assert len(offs) == nint - 1, "restart markers (%d) do not match the restart interval (%d intervals)" % (len(offs), nint)
    rows_per = (total_mcu_rows + nstripes - 1) // nstripes
    out = []
    for s in range(nstripes):
        r0 = min(s * rows_per, total_mcu_rows)
        r1 = min(r0 + rows_per, total_mcu_rows)
        if r1 <= r0:
            out.append((r0, 0, 0, 0))
            continue
        assert (r0 * mcus_per_row) % restart_interval == 0, "stripe does not start on a restart interval"
        i0 = (r0 * mcus_per_row) // restart_interval
        i1 = (r1 * mcus_per_row + restart_interval - 1) // restart_interval
        b0 = 0 if i0 == 0 else int(offs[i0 - 1]) + 2
        b1 = len(scan) if i1 >= nint else int(offs[i1 - 1])
        out.append((r0, r1 - r0, b0, b1))
    return out
