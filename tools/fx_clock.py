#!/usr/bin/env python3
"""tools/fx_clock.py -- where a wavefront of k_fixup spends its cycles (GPU box; needs build/ablate/libkpeg_hip_fxstamp.so =
-DKPEG_FX_STAMP).  Every wavefront of the diagnostic build adds up the shader cycles of its phases (s_memtime) and stores
its start and lifetime (s_memrealtime, 100 MHz) to a slot of its own."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PHASES = ["set-up", "mask+scan", "list", "row+corner", "lane blocks", "wave samples", "colour+store", "-"]


def main():
    import torch
    import libkpeg_amd as K
    torch.cuda.set_stream(torch.cuda.Stream())
    lib = K.load_variant(os.path.join(ROOT, "build", "ablate", "libkpeg_hip_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "fxstamp")))
    W, H = bench.W8K, bench.H8K
    rc, frame, scan = K.host_parse(bench.synth_jpeg(W, H))
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    d_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    ctx = K.Context(0, lib=lib)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(100):
        ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    ctx.sync()
    pct = lambda a: " ".join("%8.0f" % np.percentile(a, q) for q in (0, 10, 50, 90, 99, 100))
    for rep in range(2):
        ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
        ctx.sync()
        st = np.zeros(8192 * 16, np.uint64)
        assert lib.kpeg_hip_debug_fx_stamps(st.ctypes.data_as(ctypes.c_void_p), st.size) == 0
        st = st.reshape(-1, 16)
        st = st[st[:, 8] > 0]
        start = (st[:, 8] & ((1 << 40) - 1)).astype(np.int64)
        dur = (st[:, 8] >> 40).astype(np.int64)
        t0 = start.min()
        print("k_fixup: %d wavefronts; start %s | end %s (x 10 ns, percentiles 0 10 50 90 99 100)" % (len(st), pct(start - t0), pct(start - t0 + dur)))
        for k, name in enumerate(PHASES):
            print("   %-20s cycles %s" % (name, pct(st[:, k].astype(np.int64))))
        print("   %-20s cycles %s" % ("all phases", pct(st[:, :8].astype(np.int64).sum(1))))


if __name__ == "__main__":
    main()
