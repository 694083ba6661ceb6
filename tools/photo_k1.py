#!/usr/bin/env python3
"""tools/photo_k1.py -- K1 on photographic content: the golden photographs tiled to 3840x2160 and 7680x4320 and re-encoded
(4:4:4) at several qualities (GPU box).  Real pictures do have workgroups whose assumed entry state fails (synthetic
fields do not): K1's time and the number of its launches that had work, with the bit rate's own sub-sequence size and with
the other one forced.  (profiles/r02_g_settle_kernel_vs_three_launches_photographs.txt is this table for the serial repair
that was built and dropped.)"""
import io, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import libkpeg_amd as K, kpeg_testlib as T
from PIL import Image
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0)
ctx.set_profiling(True)
for src in ("nat_china_640x424_q90.jpg", "nat_flower_640x424_q75_opt.jpg", "lena.jpg"):
    im = np.asarray(Image.open("tests/golden/" + src).convert("RGB"))
    for (w, h) in ((3840, 2160), (7680, 4320)):
        big = np.tile(im, (h // im.shape[0] + 1, w // im.shape[1] + 1, 1))[:h, :w]
        for q in (50, 75, 85, 90, 93):
            data = T.encode_rgb(np.ascontiguousarray(big), quality=q)
            rc, f, scan = K.host_parse(data)
            bpp = len(scan) * 8 / (f.width * f.height)
            out, pix = [], []
            for subseq in (96, 384):
                assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, subseq) == 0
                ts = []
                for _ in range(4):
                    rgb = ctx.decode_scan(f, scan)
                    t = ctx.timings()
                    ts.append(t["huff_sync_ms"])
                out.append((min(ts), int(t["sync_rounds"])))
                pix.append(rgb)
            same = bool(np.array_equal(pix[0], pix[1]))
            print("%-30s %dx%d q%d %.2f bits/px  96-bit sub-sequences: K1 %.3f ms (launches with work %d)   384-bit: K1 %.3f ms (%d)  same pixels: %s"
                  % (src, w, h, q, bpp, out[0][0], out[0][1], out[1][0], out[1][1], same), flush=True)
