#!/usr/bin/env python3
"""tools/subseq_choice.py -- K1 + K2 with 96- and with 384-bit sub-sequences by bit rate, synthetic fields and tiled
photographs, 3840x2160 and 7680x4320 (GPU box): where should the host switch?"""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, libkpeg_amd as K, kpeg_testlib as T
from PIL import Image
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0)
ctx.set_profiling(True)
cases = []
for (w, h) in ((3840, 2160), (7680, 4320)):
    for (q, sigma) in ((75, 6.0), (85, 6.0), (88, 6.0), (90, 6.0), (92, 6.0), (90, 10.0)):
        cases.append(("synthetic q%d sigma%g %dx%d" % (q, sigma, w, h), bench.synth_jpeg(w, h, quality=q, sigma=sigma)))
    for src in ("nat_china_640x424_q90.jpg", "nat_flower_640x424_q75_opt.jpg", "lena.jpg"):
        im = np.asarray(Image.open("tests/golden/" + src).convert("RGB"))
        big = np.ascontiguousarray(np.tile(im, (h // im.shape[0] + 1, w // im.shape[1] + 1, 1))[:h, :w])
        for q in (50, 75, 85, 90, 93):
            cases.append(("%s q%d %dx%d" % (src.split("_")[1] if src.startswith("nat") else "lena", q, w, h), T.encode_rgb(big, quality=q)))
for name, data in cases:
    rc, f, scan = K.host_parse(data)
    bpp = len(scan) * 8 / (f.width * f.height)
    out = []
    for subseq in (96, 384):
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, subseq) == 0
        best = None
        for _ in range(4):
            ctx.decode_scan(f, scan)
            t = ctx.timings()
            v = (t["huff_sync_ms"] + t["huff_write_ms"], t["huff_sync_ms"], t["huff_write_ms"])
            best = v if best is None or v[0] < best[0] else best
        out.append(best)
    print("%-36s %.2f bits/px   96: K1+K2 %.3f (%.3f + %.3f)   384: %.3f (%.3f + %.3f)   better: %s" %
          (name, bpp, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2], "96" if out[0][0] <= out[1][0] else "384"), flush=True)
