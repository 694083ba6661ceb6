#!/usr/bin/env python3
"""Debug aid: decode one synthetic picture on the GPU, list the pixels that differ from the oracle with what their blocks look like.

    python tools/k4_mismatch.py [w h q sigma]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kpeg_testlib as T
import libkpeg_amd as K

w, h, q, sigma = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 75, 6.0)
data = T.synth_jpeg(w, h, seed=17, quality=q, sigma=sigma, mode=0)
st, want = T.oracle_decode(data)
p = T.oracle_parse(data)
rc, zz = T.oracle_entropy(p)
nat = T.zz_to_natural(zz)   # [mcu,3,8,8]
ctx = K.Context()
ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2)
got = ctx.decode_scan(T.make_frame(p, 0), p.scan)
bad = np.argwhere((got != want).any(axis=2))
print("%d pixels differ" % len(bad))
mw = w // 8
for (y, x) in bad[:40]:
    mcu = (y // 8) * mw + x // 8
    info = []
    for c in range(3):
        b = nat[mcu, c].astype(int)
        ac = b.copy(); ac[0, 0] = 0
        nnz = int((ac != 0).sum())
        oc = ac.copy(); oc[0, 1] = oc[1, 0] = oc[1, 1] = 0
        info.append("c%d nnz %d %s" % (c, nnz, "corner" if not oc.any() else "other"))
    print("(%d,%d) mcu %d tile %d grp %d row %d col %d  got %s want %s  %s" % (y, x, mcu, mcu // 8, mcu % 8, y % 8, x % 8, got[y, x].tolist(), want[y, x].tolist(), "; ".join(info)))

# what the samples of the differing pixels are (double precision, not the reference's float accumulator: shows near-ties)
import math
qt = np.asarray(p.qt, dtype=np.float64).reshape(-1, 64)[:2] if hasattr(p, "qt") else None
zzt = T.zz_table()
def sample(mcu, c, x, y):
    q = qt[1 if c else 0]
    qn = np.zeros(64); qn[zzt] = q          # quantisers to natural order (p.qt is zig-zag)
    b = nat[mcu, c].astype(np.float64) * qn.reshape(8, 8)
    s = 0.0
    for u in range(8):
        for v in range(8):
            if b[u, v]:
                cu = math.sqrt(0.5) if u == 0 else 1.0
                cv = math.sqrt(0.5) if v == 0 else 1.0
                s += cu * cv * b[u, v] * math.cos((2 * x + 1) * u * math.pi / 16) * math.cos((2 * y + 1) * v * math.pi / 16)
    return 0.25 * s
def sample_ref(mcu, c, x, y):
    """MCU::computeIDCT's arithmetic: float accumulator, double products, u outer, v inner; then roundl + 128"""
    q = qt[1 if c else 0]
    qn = np.zeros(64); qn[zzt] = q
    F = nat[mcu, c].astype(np.int64) * qn.reshape(8, 8).astype(np.int64)
    s = np.float32(0.0)
    c0 = np.float32(1.0) / np.sqrt(np.float32(2.0))
    for u in range(8):
        for v in range(8):
            if F[u, v]:
                cc = np.float32((c0 if u == 0 else np.float32(1.0)) * (c0 if v == 0 else np.float32(1.0)))
                fc = np.float32(cc * np.float32(F[u, v]))
                t = (float(fc) * math.cos((2 * x + 1) * u * math.pi / 16)) * math.cos((2 * y + 1) * v * math.pi / 16)
                s = np.float32(float(s) + t)
    ic = 0.25 * float(s)
    return int(math.floor(abs(ic) + 0.5) * (1 if ic >= 0 else -1)) + 128
def colour(Y, Cb, Cr):
    R = math.floor(Y + 1.402 * (Cr - 128.0)); G = math.floor(Y - 0.344136 * (Cb - 128.0) - 0.714136 * (Cr - 128.0)); B = math.floor(Y + 1.772 * (Cb - 128.0))
    return [max(0, min(255, int(v))) for v in (R, G, B)]
if qt is not None:
    for (y, x) in bad[:12]:
        mcu = (y // 8) * mw + x // 8
        Sr = [sample_ref(mcu, c, y % 8, x % 8) for c in range(3)]
        Sf = [int(np.rint(sample(mcu, c, y % 8, x % 8))) + 128 for c in range(3)]
        print("   ref samples", Sr, "->", colour(*Sr), " fast (rint)", Sf, "->", colour(*Sf))
        print("(%d,%d)" % (y, x), " ".join("c%d %.6f nz %s" % (c, sample(mcu, c, y % 8, x % 8), [(int(u), int(v), int(nat[mcu, c, u, v])) for u, v in np.argwhere(nat[mcu, c] != 0)]) for c in range(3)))
