"""Build recipe for the in-tree native libraries (driven by __graft_entry__.build()).

    python -m libkpeg_amd.build          # everything
    python -m libkpeg_amd.build hip      # only libkpeg_hip.so

hipcc cross-compiles for gfx950 without a GPU; the .so files stay in-tree (git-ignored)
so that they travel to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "libkpeg_amd")
CSRC = os.path.join(PKG, "csrc")

HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # bit-exactness contract: no FMA contraction anywhere; FMAs are written explicitly
    "-ffp-contract=off",
    # no SLP vectorisation: it turns pairs of f32 adds/muls/fmas into v_pk_* instructions, which issue at 2.7 cycles
    # where the scalar forms issue at 1.7 (tools/ubench/valu_rate.hip), and pays v_mov/s_mov to line operands up
    "-fno-slp-vectorize",
    "-Wall",
    "-Wno-unused-function",
]


def source_hash(sources, flags):
    """SHA-256 over the contents of `sources` (by base name, sorted) and the compiler flags."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(sources, key=os.path.basename):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    h.update(" ".join(flags).encode())
    return h.hexdigest()[:32]


def stamped_hash(target):
    """The source hash a binary was built from (it carries the string KPEG_SRC_HASH=<hex>), or None."""
    if not os.path.exists(target):
        return None
    blob = open(target, "rb").read()
    i = blob.find(b"KPEG_SRC_HASH=")
    if i < 0:
        return None
    return blob[i + 14:i + 14 + 32].decode("ascii", "replace")


def _stale(target, want):
    """Rebuild unless the binary exists and was built from exactly these sources and flags: a git-ignored .so
    left over from other sources must never travel to the GPU box (modification times say nothing after a checkout)."""
    return stamped_hash(target) != want


def hip_sources():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "kpeg_hip.h"))
    return srcs


def host_sources():
    host_dir = os.path.join(CSRC, "host")
    srcs = [os.path.join(host_dir, f) for f in sorted(os.listdir(host_dir)) if f.endswith(".cpp")]
    hdrs = [os.path.join(ROOT, "include", "kpeg", f) for f in os.listdir(os.path.join(ROOT, "include", "kpeg"))]
    return srcs + hdrs + [os.path.join(ROOT, "include", "kpeg_host.h"), os.path.join(ROOT, "include", "kpeg_hip.h")]


HOST_FLAGS = ["-O2", "-std=c++14", "-Wall"]
STRESS_DEFS = ["-DKPEG_SUBSEQ_BITS=64", "-DKPEG_SYNC_WG=128", "-DKPEG_WARM_BITS=64", "-DKPEG_POOL_SUBS=2"]


def _run(cmd, cwd=None):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=cwd)


def build_hip(force=False):
    out = os.path.join(PKG, "libkpeg_hip.so")
    stress = os.path.join(PKG, "libkpeg_hip_stress.so")
    h = source_hash(hip_sources(), HIP_FLAGS + STRESS_DEFS)
    if not force and not _stale(out, h) and not _stale(stress, h):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    stamp = ['-DKPEG_SRC_HASH="%s"' % h]
    _run([hipcc] + HIP_FLAGS + stamp + ["-o", out, os.path.join(CSRC, "kpeg_hip.hip")])
    # test-only twin with tiny K1/K2 workgroups and a 64-bit warm-up: real streams then need the boundary
    # passes and the chained pass that the product geometry almost never reaches; and a pool of two
    # second-level Huffman tables, so that the Annex-K tables overflow it and long codes take the
    # canonical-search fallback (tests/test_gpu_decode.py)
    _run([hipcc] + HIP_FLAGS + stamp + STRESS_DEFS + ["-o", stress, os.path.join(CSRC, "kpeg_hip.hip")])
    return out


def build_host(force=False):
    """kpeg::JPEGDecoder / kpeg::Image host library and the `kpeg` CLI (C++, links libkpeg_hip)."""
    host_dir = os.path.join(CSRC, "host")
    if not os.path.isdir(host_dir):
        return None
    out = os.path.join(PKG, "libkpeg.so")
    cli = os.path.join(PKG, "kpeg")
    srcs = [os.path.join(host_dir, f) for f in sorted(os.listdir(host_dir)) if f.endswith(".cpp") and f != "main.cpp"]
    h = source_hash(host_sources(), HOST_FLAGS)
    stamp = ['-DKPEG_SRC_HASH="%s"' % h]
    if force or _stale(out, h):
        _run(["g++"] + HOST_FLAGS + stamp + ["-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
              "-I" + os.path.join(ROOT, "include", "kpeg"), "-o", out] + srcs +
             ["-L" + PKG, "-lkpeg_hip", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    main = os.path.join(host_dir, "main.cpp")
    if os.path.exists(main) and (force or _stale(cli, h)):
        _run(["g++"] + HOST_FLAGS + stamp + ["-I" + os.path.join(ROOT, "include"),
              "-I" + os.path.join(ROOT, "include", "kpeg"), "-o", cli, main, "-L" + PKG, "-lkpeg", "-lkpeg_hip", "-lpthread",
              "-Wl,-rpath,$ORIGIN"])
    return out


def build_oracle(force=False):
    """The checker (CPU restatement) and, where /root/reference exists, the real reference."""
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])
    if os.path.isdir("/root/reference"):
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    _run(["make", "-s", "-C", os.path.join(ROOT, "tools"), "all"])


def build_all(force=False):
    build_hip(force)
    build_host(force)
    build_oracle(force)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    {"hip": build_hip, "host": build_host, "oracle": build_oracle, "all": build_all}[what](True)
