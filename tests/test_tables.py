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
    ent = open(os.path.join(CSRC, "entropy.hip.h")).read() + open(os.path.join(CSRC, "k2_core.inc.h")).read()
    assert "(0x1.004p-24f * Asum) * ((float)nnz + %sf)" % ("%.1f" % kk) in ent
    assert "Asum < 249.0f" in ent and "Asum < 2040.0f" in ent and "KPEG_A_LIM_CHROMA 249.0f" in src and "KPEG_A_LIM 2040.0f" in src
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
    KPEG_A_LIM_CHROMA, |luma| <= 2041 by KPEG_A_LIM; the test covers twice that) must give the reference's clamp(floor(exact value)) (MCU.cpp:259-265
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


def _luts(data, frame=None):
    import ctypes
    import libkpeg_amd
    lib = libkpeg_amd.load_hip()
    if frame is None:
        frame = T.make_frame(T.oracle_parse(data))
    bits = ctypes.c_int(0)
    lut = np.zeros((4, 512), np.uint32)
    lutx = np.zeros((4, 512), np.uint32)
    lib.kpeg_hip_debug_entropy_luts.argtypes = [ctypes.c_void_p] * 4
    assert lib.kpeg_hip_debug_entropy_luts(ctypes.byref(frame), lut.ctypes.data, lutx.ctypes.data, ctypes.byref(bits)) == 0
    assert bits.value == 9
    return lut, lutx


def test_two_symbol_entries_are_two_steps_through_the_one_symbol_table():
    """K1's exit-state decodes take symbols two at a time from lutx (entropy.hip.h, run_exit): AC AC while k < 48, DC AC,
    DC EOB.  Every such entry must be what two steps through the one-symbol tables give -- bits used, coefficient index,
    Q1 flag and table afterwards -- for every k it can be looked up with.  Host code only: no GPU."""
    E_LONG, E_BAD, E_ZERO, E_ACSYM, E_DCRUN, E_KEEP = 1 << 31, 1 << 23, 1 << 15, 1 << 27, 1 << 24, 1 << 25

    def step(e, k, q, tb):
        """one symbol step as run_count and K2 take it (entropy.hip.h): (k, q, table) afterwards"""
        kraw = k + ((e >> 16) & 127)
        if kraw >= 64:
            return (e >> 14) & 1, (e >> 25) & 1, tb + 1
        return kraw, q, tb

    def xstep(x, k, q, tb):
        """the same for an entry of lutx, as run_exit reads it"""
        kraw = k + ((x >> 16) & 127)
        if kraw >= 64:
            return (x >> 28) & 7, (x >> 25) & 1, tb + (2 if x & E_DCRUN else 1)
        return kraw, q, tb

    def random_frames(n):
        """frames whose four Huffman tables are random prefix codes (random lengths under Kraft's bound, random symbols):
        long codes in the first-level window's place, sparse tables, DC symbols with a run nibble -- what real files never have"""
        base = T.make_frame(T.oracle_parse(T.synth_jpeg(64, 64, seed=3)))
        rng = np.random.default_rng(2024)
        for _ in range(n):
            import copy
            f = copy.deepcopy(base)
            for cls in range(2):
                for tid in range(2):
                    nsym = int(rng.integers(2, 13 if cls == 0 else 163))
                    lens, budget = [], 1.0
                    for _k in range(nsym):
                        lo = 1
                        while lo <= 16 and 2.0 ** -lo > budget - (nsym - len(lens) - 1) * 2.0 ** -16:
                            lo += 1
                        if lo > 16:
                            break
                        ln = int(rng.integers(lo, min(16, lo + 6) + 1))
                        lens.append(ln)
                        budget -= 2.0 ** -ln
                    lens.sort()
                    syms = rng.permutation(256 if (cls or rng.random() < 0.3) else 16)[:len(lens)]   # (some DC tables with run nibbles)
                    d = f.dht[cls][tid]
                    for i in range(16):
                        d.counts[i] = sum(1 for ln in lens if ln == i + 1)
                    for i, sy in enumerate(syms):
                        d.symbols[i] = int(sy)
            yield f

    cases = [(_luts(T.synth_jpeg(64, 64, seed=3)), True), (_luts(open(os.path.join(T.GOLDEN, "lena.jpg"), "rb").read()), True)]
    cases += [(_luts(None, f), False) for f in random_frames(12)]
    total_pairs = 0
    for (lut, lutx), real in cases:
        pairs = dcpairs = 0
        for tid in range(2):
            dc1, ac1 = lut[tid], lut[2 + tid]
            for j in range(512):
                # AC AC
                e1, x = int(ac1[j]), int(lutx[2 + tid][j])
                if x != e1:
                    pairs += 1
                    # only an AC symbol that is neither long, nor missing, nor EOB opens a pair
                    assert not (e1 & (E_LONG | E_BAD | E_ZERO)) and (e1 & E_ACSYM)
                    len1 = e1 & 31
                    e2 = int(ac1[(j << len1) & 511])
                    assert not (e2 & (E_LONG | E_BAD))
                    assert len1 + ((e2 >> 5) & 31) <= 9      # the second code lies inside the window: the entry is the same for all its copies
                    assert (x & 31) == len1 + (e2 & 31) <= 31
                    assert 1 <= (e1 >> 16) & 127 <= 16
                    for k in range(1, 48):
                        for q in (0, 1):
                            k1, q1, t1 = step(e1, k, q, 1)
                            assert t1 == 1               # (the first symbol cannot end the block below k = 48)
                            assert step(e2, k1, q1, t1) == xstep(x, k, q, 1)
                else:
                    # a one-symbol entry read as run_exit reads it
                    if not (e1 & E_LONG):
                        for k in (1, 20, 47, 48, 63):
                            assert step(e1, k, 1, 1) == xstep(x, k, 1, 1)
                # DC AC, DC EOB (k = 0 at a DC table, always)
                d1, x = int(dc1[j]), int(lutx[tid][j])
                if d1 & E_LONG:
                    assert x == d1
                    continue
                if x == d1 & ~E_DCRUN:
                    assert step(d1, 0, 0, 0) == xstep(x, 0, 0, 0)
                    continue
                dcpairs += 1
                assert not (d1 & (E_BAD | E_DCRUN))
                len1 = d1 & 31
                e2 = int(ac1[(j << len1) & 511])
                assert not (e2 & (E_LONG | E_BAD)) and len1 + ((e2 >> 5) & 31) <= 9
                assert (x & 31) == len1 + (e2 & 31) <= 31
                for q in (0, 1):
                    k1, q1, t1 = step(d1, 0, q, 0)
                    assert (k1, t1) == (1, 1) and q1 == (1 if d1 & E_KEEP else 0)
                    assert step(e2, k1, q1, t1) == xstep(x, 0, q, 0)
        assert not real or (pairs > 100 and dcpairs > 50), (pairs, dcpairs)
        total_pairs += pairs + dcpairs
    assert total_pairs > 2000, total_pairs   # (the random tables pair as well)
