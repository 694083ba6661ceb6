#!/usr/bin/env python3
"""tools/k4_clock.py -- per-wavefront timeline of K4 (GPU box; needs build/ablate/libkpeg_hip_stamp.so = -DKPEG_K4_STAMP).
Every wavefront of the diagnostic build stores its start and end (s_memrealtime, 100 MHz), its lifetime in shader
cycles (s_memtime) and where it ran (XCC_ID, HW_ID) to a slot of its own.  Prints the shader clock K4 runs at (the
ratio of the two clocks: MI355X_MICROARCH.md, DVFS give-back, item 6) and how evenly the wavefronts finish."""
import collections
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import libkpeg_amd as K
    torch.cuda.set_stream(torch.cuda.Stream())   # a created stream: the default stream's handle 0 cannot be handed to the C ABI
    lib = K.load_variant(os.path.join(ROOT, "build", "ablate", "libkpeg_hip_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamp")))
    W, H = bench.W8K, bench.H8K
    rc, frame, scan = K.host_parse(bench.synth_jpeg(W, H))
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    d_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    ctx = K.Context(0, lib=lib)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if os.environ.get("KPEG_IDCT_MODE"):
        ctx.set_idct_mode(int(os.environ["KPEG_IDCT_MODE"]))   # 2: marked pixels are counted, not settled (wrong pixels; what the tile loop alone does)
    for _ in range(200):   # a few ms of back-to-back launches before the measured one
        ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    ctx.sync()
    pct = lambda a: " ".join("%.1f" % np.percentile(a, q) for q in (0, 10, 50, 90, 99, 100))
    for rep in range(3):
        ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
        ctx.sync()
        st = np.zeros(8192 * 8, np.uint64)
        assert lib.kpeg_hip_debug_k4_stamps(st.ctypes.data_as(ctypes.c_void_p), st.size) == 0
        ex = st[8192 * 4:].reshape(-1, 4).astype(np.int64)   # tile loop done, cycles in fx_flush | calls << 48, cycles in the non-corner block | times << 48
        st = st[:8192 * 4].reshape(-1, 4).astype(np.int64)
        ex = ex[st[:, 1] > 0]
        st = st[st[:, 1] > 0]
        nwaves = len(st)
        t0 = st[:, 0].min()
        start, end, life = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, (st[:, 1] - st[:, 0]) / 100.0
        unsafe = (st[:, 3] >> 48) & 0xFFFF
        print("K4: %d wavefronts, mean lifetime %.1f us, shader clock %.0f MHz" % (nwaves, life.mean(), st[:, 2].sum() / (st[:, 1] - st[:, 0]).sum() * 100.0))
        print("   us, percentiles 0 10 50 90 99 100: start %s | end %s | lifetime %s" % (pct(start), pct(end), pct(life)))
        if ex[:, 0].max() > 0:
            clk = st[:, 2].sum() / (st[:, 1] - st[:, 0]).sum() * 100.0   # MHz
            m48 = (1 << 48) - 1
            tail = (st[:, 1] - ex[:, 0]) / 100.0
            fl_us, fl_n = (ex[:, 1] & m48) / clk, ex[:, 1] >> 48
            nc_us, nc_n = (ex[:, 2] & m48) / clk, ex[:, 2] >> 48
            print("   tile loop done at %s | from there to the end %s" % (pct((ex[:, 0] - t0) / 100.0), pct(tail)))
            print("   fx_flush: calls per wave %s, us per wave %s | non-corner block: times %s, us %s" % (pct(fl_n), pct(fl_us), pct(nc_n), pct(nc_us)))
        if rep:
            continue
        xcc, hw = (st[:, 3] >> 32) & 15, st[:, 3] & 0xFFFFFFFF
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        percu = collections.Counter(cuid.tolist())
        persimd = collections.Counter((cuid * 4 + simd).tolist())
        print("   distinct CUs %d, waves per CU %s, per SIMD %s" % (len(percu), sorted(collections.Counter(percu.values()).items()),
                                                                      sorted(collections.Counter(persimd.values()).items())))
        cu_end = collections.defaultdict(float)
        for c, e in zip(cuid.tolist(), end.tolist()):
            cu_end[c] = max(cu_end[c], e)
        print("   last end per CU (us): %s" % pct(np.array(list(cu_end.values()))))
        print("   end median by xcc: %s" % " ".join("%d:%.1f" % (k, np.median(end[xcc == k])) for k in sorted(set(xcc.tolist()))))
        print("   unsafe pixels per wave median %d max %d; corr(lifetime, unsafe) %.2f" % (np.median(unsafe), unsafe.max(), np.corrcoef(life, unsafe)[0, 1]))


if __name__ == "__main__":
    main()
