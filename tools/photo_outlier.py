#!/usr/bin/env python3
"""tools/photo_outlier.py -- the one input of profiles/r02_h_fused_on_photographs.txt whose separate-launch time stood out (lena tiled to
3840x2176, q75: 0.49 ms against 0.14 ms with k_sync_write, neighbours within 3 %): both paths again, five rounds each, with the
library's own per-kernel events (GPU box)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import libkpeg_amd as K, kpeg_testlib as T
from PIL import Image
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0)
assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0
im = np.asarray(Image.open("tests/golden/lena.jpg").convert("RGB"))
for (w, h, q) in ((3840, 2176, 75), (3840, 2176, 50), (7680, 4352, 75)):
    big = np.ascontiguousarray(np.tile(im, (h // im.shape[0] + 1, w // im.shape[1] + 1, 1))[:h, :w])
    data = T.encode_rgb(big, quality=q)
    rc, f, scan = K.host_parse(data)
    st, want = T.oracle_decode(data)
    d_scan = torch.frombuffer(bytearray(scan), dtype=torch.uint8).cuda()
    d_rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    for fused in (0, 1):
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, fused) == 0
        ok = bool(np.array_equal(ctx.decode_scan(f, scan), want))
        walls = []
        for r in range(5):
            ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
            ctx.sync(); torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) / 50 * 1e3)
        ctx.set_profiling(True)
        acc = {}
        for _ in range(5):
            ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync()
            for k, v in ctx.timings().items(): acc[k] = acc.get(k, 0.0) + v / 5
        ctx.set_profiling(False)
        print("lena tiled %dx%d q%d %.2f bits/px  k_sync_write %d  pixels %s  wall ms %s  | K1 launches with work %d  events: K1 %.4f (verify+chained %.4f) K2 %.4f K4 %.4f" % (
            w, h, q, len(scan) * 8 / (w * h), fused, "ok" if ok else "WRONG", " ".join("%.4f" % x for x in walls), acc["sync_rounds"],
            acc["huff_sync_ms"], acc["huff_scan_ms"], acc["huff_write_ms"], acc["idct_ms"]), flush=True)
