"""The C-ABI libraries load and export every entry point their headers declare (no compute calls:
this runs without a GPU)."""
import ctypes
import os
import re

import kpeg_testlib as T

ROOT = T.ROOT


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kpeg_(?:hip|host)_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports_every_declared_symbol():
    import libkpeg_amd
    names = _declared("kpeg_hip.h")
    assert len(names) >= 17
    lib = libkpeg_amd.load_hip()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(libkpeg_amd.EXPORTS) == [n for n in names if n in libkpeg_amd.EXPORTS]
    assert set(libkpeg_amd.EXPORTS) == set(names), set(names) ^ set(libkpeg_amd.EXPORTS)
    assert lib.kpeg_hip_abi_version() == libkpeg_amd.ABI_VERSION
    assert lib.kpeg_hip_strerror(-4) == b"corrupt or truncated entropy-coded data"


def test_host_library_exports_every_declared_symbol():
    import libkpeg_amd
    lib = libkpeg_amd.load_host()
    for n in _declared("kpeg_host.h"):
        assert hasattr(lib, n), n


def test_frame_struct_layout_matches_header():
    import libkpeg_amd
    # uint32 x2, uint16[2][64], {uint8[16], uint8[256]}[2][2], uint32
    assert ctypes.sizeof(libkpeg_amd.Dht) == 272
    assert ctypes.sizeof(libkpeg_amd.Frame) == 8 + 256 + 4 * 272 + 8
    assert libkpeg_amd.Frame.restart_interval.offset == 8 + 256 + 4 * 272
    assert libkpeg_amd.Frame.components.offset == 8 + 256 + 4 * 272 + 4


def test_no_cpu_fallback_without_gpu():
    """On a machine without a gfx950 device context creation must fail loudly, not fall back."""
    import libkpeg_amd
    import torch
    if torch.cuda.is_available():
        return
    try:
        libkpeg_amd.Context(0)
    except libkpeg_amd.KpegError as e:
        assert e.code == libkpeg_amd.E_DEVICE
    else:
        raise AssertionError("Context() succeeded without a GPU")


def test_product_does_not_link_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/: the shipped libraries must not."""
    for lib in ("libkpeg_hip.so", "libkpeg.so", "kpeg"):
        path = os.path.join(ROOT, "libkpeg_amd", lib)
        blob = open(path, "rb").read()
        assert b"kpeg_oracle" not in blob and b"libkpeg_synth" not in blob, lib
    for root, _, files in os.walk(os.path.join(ROOT, "libkpeg_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "kpeg_oracle" not in txt and "oracle/" not in txt.replace("oracle/)", ""), os.path.join(root, f)


def test_binaries_carry_the_hash_of_the_sources_in_the_tree():
    """A stale git-ignored .so must never reach the GPU box: every shipped binary is stamped with the hash of
    the sources and flags it was built from, and build() rebuilds on a mismatch (libkpeg_amd/build.py)."""
    import libkpeg_amd
    from libkpeg_amd import build as B
    hip = B.source_hash(B.hip_sources(), B.HIP_FLAGS + B.STRESS_DEFS)
    host = B.source_hash(B.host_sources(), B.HOST_FLAGS)
    pkg = os.path.join(ROOT, "libkpeg_amd")
    assert B.stamped_hash(os.path.join(pkg, "libkpeg_hip.so")) == hip
    assert B.stamped_hash(os.path.join(pkg, "libkpeg_hip_stress.so")) == hip
    assert B.stamped_hash(os.path.join(pkg, "libkpeg.so")) == host
    assert B.stamped_hash(os.path.join(pkg, "kpeg")) == host
    assert libkpeg_amd.load_hip().kpeg_hip_build_hash().decode() == hip
    H = libkpeg_amd.load_host()
    H.kpeg_host_build_hash.restype = ctypes.c_char_p
    assert H.kpeg_host_build_hash().decode() == host
