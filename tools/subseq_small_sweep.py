#!/usr/bin/env python3
"""tools/subseq_small_sweep.py -- where the 64-bit sub-sequences (SUBSEQ_SMALL) pay: synthetic fields and tiled photographs of growing
size, decoded with the sub-sequence size forced (debug key 4) to 64 and to 96 bits and with the library's own choice (0).  GPU box."""
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
    cases = [("synthetic %dx%d q75" % (w, h), bench.synth_jpeg(w, h)) for w, h in ((512, 512), (1920, 1088), (2560, 1440), (3840, 2160), (5120, 2880), (7680, 4320))]
    for src, q, w, h in (("lena.jpg", 50, 1024, 1024), ("lena.jpg", 75, 1024, 1024), ("lena.jpg", 50, 2048, 2048), ("lena.jpg", 75, 2048, 2048),
                         ("nat_china_640x424_q90.jpg", 75, 1920, 1088), ("nat_china_640x424_q90.jpg", 60, 2560, 1472)):
        cases.append(("%s q%d tiled %dx%d" % (src, q, w, h), bench.tiled_photo_jpeg(src, q, w, h)))
    print("%-44s %8s %9s  ms per picture with 64 / 96 / the library's choice" % ("picture", "bits/px", "scan MB"))
    for name, data in cases:
        rc, frame, scan = K.host_parse(data)
        assert rc == K.DECODE_DONE, name
        d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
        d_rgb = torch.zeros((frame.height, frame.width, 3), dtype=torch.uint8, device="cuda")
        ref = None
        out = []
        for ss in (64, 96, 0):
            assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, ss) == 0
            ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
            ctx.sync()
            got = d_rgb.cpu().numpy()
            if ref is None:
                ref = got.copy()
            assert np.array_equal(got, ref), (name, ss)
            best = 1e9
            for rep in range(5):
                for _ in range(10):
                    ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
                ctx.sync()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
                ctx.sync()
            out.append(best)
        print("%-44s %8.2f %9.2f  %.4f / %.4f / %.4f" % (name, len(scan) * 8 / (frame.width * frame.height), len(scan) / 1e6, out[0], out[1], out[2]), flush=True)
    ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)


if __name__ == "__main__":
    main()
