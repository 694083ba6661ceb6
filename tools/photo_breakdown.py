#!/usr/bin/env python3
"""tools/photo_breakdown.py -- where the time of the bench's photographs goes (GPU box): per-kernel HIP-event times of one
decode of each of bench.PHOTO_CASES at 7680x4352, with the bit rate's own sub-sequence size and with the others forced."""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
import libkpeg_amd as K

torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0)
ctx.set_profiling(True)
import os
if os.environ.get("KPEG_IDCT_MODE"):
    ctx.set_idct_mode(int(os.environ["KPEG_IDCT_MODE"]))   # 2: marked pixels are counted, not settled (wrong pixels: what K4's tile loop alone costs)
sizes = [int(a) for a in sys.argv[1:]] or [0]
cases = list(bench.PHOTO_CASES)
if os.environ.get("KPEG_MORE_PHOTOS"):   # the bit rates between the bench's cases
    cases += [("lena.jpg", 85), ("lena.jpg", 90), ("lena.jpg", 93), ("nat_china_640x424_q90.jpg", 85), ("nat_china_640x424_q90.jpg", 93)]
for (src, q) in cases:
    data = bench.tiled_photo_jpeg(src, q)
    rc, f, scan = K.host_parse(data)
    bpp = len(scan) * 8 / (f.width * f.height)
    for sb in sizes:
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, sb) == 0
        best = None
        for _ in range(5):
            ctx.decode_scan(f, scan)
            t = ctx.timings()
            if best is None or t["total_ms"] < best["total_ms"]:
                best = dict(t)
        print("%-28s q%d %.2f bits/px  sub-sequence %3s: %s" % (src, q, bpp, sb or "own", "  ".join("%s=%.4f" % (k, v) if isinstance(v, float) else "%s=%s" % (k, v) for k, v in best.items())), flush=True)
