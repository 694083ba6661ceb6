#!/usr/bin/env python3
"""tools/fused_ab.py -- k_sync_write (K1's pass 0 and K2 in one kernel, debug key 9) against the separate launches: same
pixels? wall time per decode, interleaved rounds (GPU box)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, libkpeg_amd as K
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = K.Context(0)
sizes = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or [(1920, 1088), (3840, 2176), (7680, 4320)]
for (w, h) in sizes:
    data = bench.synth_jpeg(w, h)
    rc, f, scan = K.host_parse(data)
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0   # compact stream wherever possible
    pix = []
    for fused in (0, 1):
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, fused) == 0
        pix.append(ctx.decode_scan(f, scan))
        print("  fused", fused, "launches of K1 with work:", ctx.timings().get("sync_rounds"), flush=True)
    same = bool(np.array_equal(pix[0], pix[1]))
    d_scan = torch.frombuffer(bytearray(scan), dtype=torch.uint8).cuda()
    d_rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    def loop(n):
        ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
        ctx.sync(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    res = {0: [], 1: []}
    for r in range(5):
        for fused in (0, 1):
            assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, fused) == 0
            res[fused].append(loop(200))
    same2 = bool(np.array_equal(d_rgb.cpu().numpy(), pix[0]))
    print("%dx%d  same pixels: %s / %s   separate %.4f ms   fused %.4f ms" % (w, h, same, same2, float(np.median(res[0])), float(np.median(res[1]))), flush=True)
