#!/usr/bin/env python3
"""tools/fused_photo.py -- what k_sync_write costs where it must hand on: golden photographs tiled to 3840x2176 / 7680x4352,
re-encoded; wall time per decode with debug key 9 off / on (GPU box)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import libkpeg_amd as K, kpeg_testlib as T
from PIL import Image
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0)
assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0
for src in ("nat_china_640x424_q90.jpg", "lena.jpg"):
    im = np.asarray(Image.open("tests/golden/" + src).convert("RGB"))
    for (w, h) in ((3840, 2176), (7680, 4352)):
        big = np.ascontiguousarray(np.tile(im, (h // im.shape[0] + 1, w // im.shape[1] + 1, 1))[:h, :w])
        for q in (50, 75):
            data = T.encode_rgb(big, quality=q)
            rc, f, scan = K.host_parse(data)
            d_scan = torch.frombuffer(bytearray(scan), dtype=torch.uint8).cuda()
            d_rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
            res = {}
            for fused in (0, 1):
                assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, fused) == 0
                ctx.decode_scan(f, scan); rounds = int(ctx.timings()["sync_rounds"])
                ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
                ctx.sync(); torch.cuda.synchronize()
                res[fused] = ((time.perf_counter() - t0) / 50 * 1e3, rounds)
            print("%-28s %dx%d q%d %.2f bits/px   separate %.4f ms (launches with work %d)   fused %.4f ms (%d)" %
                  (src, w, h, q, len(scan) * 8 / (w * h), res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
