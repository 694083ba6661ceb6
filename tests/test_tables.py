"""Constant tables and error-bound constants (CPU only)."""
import os
import re
import subprocess
import sys

import numpy as np

import kpeg_testlib as T

CSRC = os.path.join(T.ROOT, "libkpeg_amd", "csrc")


def _hex_doubles(text):
    return [float.fromhex(x) for x in re.findall(r"-?0x1\.[0-9a-f]+p[+-]\d+", text)]


def test_cos_table_matches_libm_and_golden():
    import ctypes
    t = (ctypes.c_double * 64)()
    T.oracle().kpeg_oracle_cos_table(t)
    libm = list(t)
    golden = [float.fromhex(l) for l in open(os.path.join(T.GOLDEN, "cos_table.hex")).read().split()]
    assert libm == golden  # this host's glibc agrees with the values captured next to the reference
    hdr = open(os.path.join(CSRC, "kpeg_tables.h")).read()
    assert _hex_doubles(hdr.split("KPEG_COS_TABLE")[1].split("};")[0]) == golden
    dev = open(os.path.join(CSRC, "idct_colour.hip.h")).read()
    assert _hex_doubles(dev.split("c_cos[64]")[1].split("};")[0]) == golden


def test_zigzag_table():
    std = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    assert list(T.zz_table()) == std
    hdr = open(os.path.join(CSRC, "kpeg_tables.h")).read()
    nums = [int(x) for x in re.findall(r"\d+", hdr.split("KPEG_ZZ_TO_NATURAL[64] = {")[1].split("}")[0])]
    assert nums == std


def test_kernel_error_constant_covers_the_derived_bound():
    """K4's KPEG_KAPPA must be >= the constant tools/idct_bound.py derives from the kernel's own
    operation sequence (which the script also checks against the exact IDCT kernel)."""
    sys.path.insert(0, os.path.join(T.ROOT, "tools"))
    import idct_bound
    k = idct_bound.kappa()
    src = open(os.path.join(CSRC, "idct_colour.hip.h")).read()
    kk = float(re.search(r"#define KPEG_KAPPA ([0-9.]+)f", src).group(1))
    assert 10.0 < k <= kk < k + 1.0, (k, kk)
    # K2 writes the same bound formula as block_ebound()
    ent = open(os.path.join(CSRC, "entropy.hip.h")).read()
    assert "(0x1.004p-24f * Asum) * ((float)nnz + %sf)" % ("%.1f" % kk) in ent
    assert "Asum < 249.0f" in ent and "Asum < 4000.0f" in ent and "KPEG_A_LIM_CHROMA 249.0f" in src and "KPEG_A_LIM 4000.0f" in src
    assert "(__float_as_uint(E) + 1u) & ~1u" in ent and "(__float_as_uint(E) + 1u) & ~1u" in src   # same flag encoding on both sides
    assert "#define KPEG_U 0x1.004p-24f" in src


def test_fast_path_error_bound_holds_empirically():
    """float32 emulation of K4's row/column passes vs the oracle's float result: the observed
    distance stays inside E = U * A * (nnz_ac + KAPPA) on real blocks."""
    data = T.synth_jpeg(256, 128, seed=8)
    p = T.oracle_parse(data)
    rc, coef = T.oracle_entropy(p)
    nat = T.zz_to_natural(coef)
    zz = T.zz_table()
    import ctypes
    import math
    C = np.array([[math.cos((2 * x + 1) * u * math.pi / 16) for u in range(8)] for x in range(8)], np.float32)
    cc = np.ones((8, 8), np.float32)
    cc[0, :] *= np.float32(float.fromhex('0x1.6a09e6p-1'))
    cc[:, 0] *= np.float32(float.fromhex('0x1.6a09e6p-1'))
    worst = 0.0
    out = np.empty(64, np.float32)
    for c in range(3):
        q = np.zeros(64, np.float32)
        q[zz] = p.qt[0 if c == 0 else 1].astype(np.float32)
        q = q.reshape(8, 8)
        for n in range(0, nat.shape[0], 7):
            blk = nat[n, c].astype(np.float32)
            inn = (np.float32(0.25) * cc * q) * blk
            inn[:, 0] = np.float32(0.25) * (cc[:, 0] * (blk[:, 0] * q[:, 0]))
            fast = (C @ inn.astype(np.float32) @ C.T).astype(np.float32)  # f32 matrix form of the same transform
            T.oracle().kpeg_oracle_idct_block(np.ascontiguousarray(coef[n, c]).ctypes.data,
                                              np.ascontiguousarray(p.qt[0 if c == 0 else 1]).ctypes.data, out.ctypes.data)
            A = float(np.abs(inn).sum())
            nn = int((blk != 0).sum() - (blk[0, 0] != 0))
            E = 2.0 ** -24 * A * (nn + 14.5) if nn else 0.0
            err = float(np.abs(fast.reshape(-1).astype(np.float64) - out.astype(np.float64)).max())
            if E > 0:
                worst = max(worst, err / E)
            else:
                assert err == 0.0
    assert worst < 1.0, worst


def test_colour_arithmetic_is_exact_over_its_whole_range():
    """K4 hands v_cvt_pk_u8_f32 (round to nearest even, saturating) the values fma(Cr', 1.402f, Yo), fma(Cb', 1.772f, Yo)
    and Yo - ceil(t) with Yo = rounded luma + 127.501f.  Every case the fast path admits (|chroma| <= 249 by
    KPEG_A_LIM_CHROMA, |luma| <= 4000 by KPEG_A_LIM) must give the reference's clamp(floor(exact value)) (MCU.cpp:259-265
    evaluates in double: exact unless the value is an integer, which happens only for chroma 0)."""
    f32 = np.float32
    ry = np.arange(-4100, 4101, dtype=np.float64)
    yo = (ry + np.float64(f32(127.501))).astype(f32)   # the sum is exact in double: one f32 rounding
    ch = np.arange(-249, 250, dtype=np.float64)

    def cvt(x):
        return np.clip(np.rint(x.astype(np.float64)), 0, 255).astype(np.int64)

    for c, num, den in ((1.402, 701, 500), (1.772, 443, 250)):
        arg = (ch[None, :] * np.float64(f32(c)) + yo[:, None].astype(np.float64)).astype(f32)   # exact in double, one rounding = fma
        exact = np.array([(int(v) * num) // den for v in ch], dtype=np.int64)   # floor(ch * c), rational arithmetic
        want = np.clip(ry[:, None].astype(np.int64) + 128 + exact[None, :], 0, 255)
        assert np.array_equal(cvt(arg), want)
    k = np.arange(-270, 271, dtype=np.float64)   # ceil(t)
    arg = (yo[:, None].astype(np.float64) - k[None, :]).astype(f32)
    assert np.array_equal(cvt(arg), np.clip(ry[:, None].astype(np.int64) + 128 - k[None, :].astype(np.int64), 0, 255))


def _luts(data):
    import ctypes
    import libkpeg_amd
    lib = libkpeg_amd.load_hip()
    p = T.oracle_parse(data)
    frame = T.make_frame(p)
    bits = ctypes.c_int(0)
    lut = np.zeros((4, 512), np.uint32)
    lutx = np.zeros((2, 512), np.uint32)
    lib.kpeg_hip_debug_entropy_luts.argtypes = [ctypes.c_void_p] * 4
    assert lib.kpeg_hip_debug_entropy_luts(ctypes.byref(frame), lut.ctypes.data, lutx.ctypes.data, ctypes.byref(bits)) == 0
    assert bits.value == 9
    return lut, lutx


def test_two_symbol_entries_are_two_steps_through_the_one_symbol_table():
    """K1's exit-state decodes take AC symbols two at a time from lutx (entropy.hip.h, run_exit) while k < 48.  Every such entry
    must be what two steps through the one-symbol table give: bits used, coefficient advance, and -- wherever the first symbol
    alone cannot end the block (k < 48) -- the same (k, table switch) afterwards, for every k.  Host code only: no GPU."""
    E_LONG, E_BAD, E_ZERO, E_ACSYM = 1 << 31, 1 << 23, 1 << 15, 1 << 27
    for data in (T.synth_jpeg(64, 64, seed=3), open(os.path.join(T.GOLDEN, "lena.jpg"), "rb").read()):
        lut, lutx = _luts(data)
        pairs = 0
        for tid in range(2):
            one = lut[2 + tid]
            for j in range(512):
                e1, x = int(one[j]), int(lutx[tid][j])
                if x == e1:
                    continue
                pairs += 1
                # only an AC symbol that is neither long, nor missing, nor EOB opens a pair
                assert not (e1 & (E_LONG | E_BAD | E_ZERO)) and (e1 & E_ACSYM)
                len1 = e1 & 31
                e2 = int(one[(j << len1) & 511])
                assert not (e2 & (E_LONG | E_BAD))
                assert len1 + ((e2 >> 5) & 31) <= 9          # the second code lies inside the window: the entry is the same for all its copies
                assert (x & 31) == len1 + (e2 & 31) <= 31
                adv1, adv2 = (e1 >> 16) & 127, (e2 >> 16) & 127
                assert 1 <= adv1 <= 16
                assert (x >> 16) & 127 == adv1 + adv2
                assert x & ~((127 << 16) | 31) == 0         # after the block: k = 0, q = 0; no other flag
                for k in range(1, 48):
                    # two steps, as run_count / K2 take them
                    k1 = k + adv1
                    assert k1 < 64
                    k2 = k1 + adv2
                    step2 = (0, True) if k2 >= 64 else (k2, False)
                    kx = k + ((x >> 16) & 127)
                    stepx = (0, True) if kx >= 64 else (kx, False)
                    assert step2 == stepx
        assert pairs > 100, pairs
