#!/usr/bin/env python3
"""tools/small_images_layout.py -- small pictures (the reference's lena.jpg, the committed photographs, synthetic fields up to 1080p):
the dense coefficient layout with separate launches against the compact stream, which lets K1's pass 0 and K2 run as one kernel
(debug key 7: 0 = the library's choice, 1 = dense, 2 = compact wherever possible).  GPU box.  Prints ms per picture, back to back."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import kpeg_testlib as T  # noqa: E402


def main():
    import torch
    import libkpeg_amd as K
    torch.cuda.set_stream(torch.cuda.Stream())
    ctx = K.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    cases = []
    for name in ("lena.jpg", "nat_china_640x424_q50.jpg", "nat_china_640x424_q90.jpg", "nat_flower_320x208_q96.jpg", "nat_flower_640x424_q75_opt.jpg"):
        cases.append((name, open(os.path.join(T.GOLDEN, name), "rb").read()))
    for w, h in ((512, 512), (1024, 768), (1920, 1088), (2560, 1440)):
        cases.append(("synthetic %dx%d q75" % (w, h), bench.synth_jpeg(w, h)))
    print("%-34s %9s  %s" % ("picture", "bits/px", "ms per picture: layout 0 (library's choice) / 1 (dense) / 2 (compact)   k1 launches with work"))
    for name, data in cases:
        rc, frame, scan = K.host_parse(data)
        assert rc == K.DECODE_DONE, name
        st, want = T.oracle_decode(data)
        d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
        d_rgb = torch.zeros((frame.height, frame.width, 3), dtype=torch.uint8, device="cuda")
        out, launches = [], []
        for layout in (0, 1, 2):
            assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
            d_rgb.zero_()
            ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
            ctx.sync()
            assert np.array_equal(d_rgb.cpu().numpy(), want), (name, layout)
            best = 1e9
            for rep in range(5):
                for _ in range(20):
                    ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
                ctx.sync()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
                ctx.sync()
            out.append(best)
            launches.append(ctx.timings().get("sync_rounds"))
        print("%-34s %9.2f  %.4f / %.4f / %.4f   %s" % (name, len(scan) * 8 / (frame.width * frame.height), out[0], out[1], out[2], launches), flush=True)
    ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


if __name__ == "__main__":
    main()
